"""Sphere / interval flow layers of the reference restated in numpy (oracle = test infrastructure only):
  'r'  layers/intervals/rational_quadratic_spline.py     'o'  layers/spheres/splines_1d.py
  'm'  layers/spheres/moebius_1d.py                      'f'  layers/spheres/fvm_2d.py
  'v'  layers/spheres/exponential_map_s2.py  (+ bisection_n_newton.py:137-256, 394-465)
plus the sphere_base / interval_base wrappers (rotation, charts) around them.
Each layer object offers  inverse(x, log_det, params) / forward(x, log_det, params) -> (x, log_det, bins)
with params a (1|B, P) row block in the reference's extra_inputs layout."""
import numpy as np

from . import manifolds as mf
from . import splines
from .special import bounded_log_fn, logsumexp, matvec

PI = np.pi
TWO_PI = 2.0 * np.pi


# =====================================================================================  spline parameter unpacking
class _SplineParamMixin:
    """width/height/derivative bookkeeping shared by 'r' and 'o' (rational_quadratic_spline.py:99-170, splines_1d.py:39-101)."""

    def _setup_counts(self, o, circular):
        nb = o["num_basis_functions"]
        self.nb = nb
        self.fix_first = o["fix_first_width_n_height_to_zero"]
        self.fix_second = o["also_fix_second_width_to_zero"]
        self.independent = o["independent_width_height_parametrization"]
        self.smooth = o["smooth_second_derivative"]
        self.fix_bd = o["fix_boundary_derivatives"]
        self.min_w, self.min_h, self.min_d = o["min_width"], o["min_height"], o["min_derivative"]
        self.n_w = self.n_h = nb
        if self.fix_first:
            self.n_w = self.n_h = nb - 1
            if self.fix_second > 0:
                self.n_w -= 1
        sub = 0
        self.bd_fixed_value = None
        if not circular:
            if self.fix_bd > 0.0:
                self.bd_fixed_value = np.log(np.exp(self.fix_bd - self.min_d) - 1.0)
            if self.smooth == 1:
                assert nb in (2, 3)
                sub = {2: (3 if self.fix_bd > 0 else 1), 3: (4 if self.fix_bd > 0 else 2)}[nb]
            elif self.fix_bd > 0.0:
                sub = 2
        else:
            if self.smooth == 1:
                assert nb == 2
                sub = 3
            elif self.fix_bd > 0.0:
                sub = 2
                self.bd_fixed_value = np.log(np.exp(self.fix_bd - self.min_d) - 1.0)
            else:
                sub = 1
        self.n_d = nb + 1 - sub
        if self.smooth and nb == 3:
            self.n_w -= 1
            self.n_h -= 1
        self.n_spline = self.n_w + self.n_h + self.n_d

    def _unpack_whd(self, p):
        w = p[:, :self.n_w]
        h = p[:, self.n_w:self.n_w + self.n_h]
        d = p[:, self.n_w + self.n_h:self.n_w + self.n_h + self.n_d] if self.n_d > 0 else None
        if self.fix_first:
            z = np.zeros_like(h[:, 0:1])
            h = np.concatenate([z, h], axis=1)
            w = np.concatenate([z, z, w] if self.fix_second else [z, w], axis=1)
        if self.independent:
            h = w + h
        return w, h, d

    def _row_from_state(self, sd, prefix):
        parts = [sd[prefix + "rel_log_widths"].reshape(-1), sd[prefix + "rel_log_heights"].reshape(-1)]
        if self.n_d > 0:
            parts.append(sd[prefix + "rel_log_derivatives"].reshape(-1))
        return np.concatenate(parts)


# =====================================================================================  'r'
class RLayer(_SplineParamMixin):
    def __init__(self, dimension, o, first, lo, hi):
        assert dimension == 1
        self._setup_counts(o, circular=False)
        self.ratio = o["restrict_max_min_width_height_ratio"]
        self.first, self.lo, self.hi = first, lo, hi
        self.total_param_num = self.n_spline

    def row_from_state(self, sd, prefix):
        return self._row_from_state(sd, prefix)[None, :]

    def _spline(self, x, params, inverse):
        x = np.clip(x, -1.0, 1.0)                                   # rational_quadratic_spline.py:185-186 / 295-296
        w, h, d = self._unpack_whd(params)
        if self.smooth == 1 and self.nb == 3:
            w = np.concatenate([w, w[:, 0:1]], axis=1)
            h = np.concatenate([h, h[:, 0:1]], axis=1)
        if self.smooth == 0:
            if self.fix_bd > 0:
                fl = np.ones(w.shape[:-1] + (1,)) * self.bd_fixed_value
                d = np.concatenate([fl, fl] if d is None else [fl, d, fl], axis=-1)      # one bin: both derivatives are the fixed ones
            y, lad, b = splines.rqs_plain(x, w, h, d, inverse, self.lo, self.hi, self.lo, self.hi,
                                          self.min_w, self.min_h, self.min_d, self.ratio)
        else:
            bd = np.ones(w.shape[:-1] + (2,)) * self.bd_fixed_value if self.fix_bd > 0 else d
            y, lad, b = splines.rqs_smooth(x, w, h, bd, inverse, self.lo, self.hi, self.lo, self.hi,
                                           self.min_w, self.min_h, self.min_d, self.ratio)
        return np.clip(y, -1.0, 1.0), lad.sum(axis=-1), [b]

    def inverse(self, x, log_det, params):
        """interval_base.inv_flow_mapping (interval_base.py:61-69) around _inv_flow_mapping (rational_quadratic_spline.py:290-400)."""
        y, lad, b = self._spline(x, params, True)
        log_det = log_det + lad
        if self.first:
            y, log_det = mf.interval_to_real_line(y, log_det, self.lo, self.hi)
        return y, log_det, b

    def forward(self, x, log_det, params):
        """interval_base.flow_mapping (:71-79) + _flow_mapping (rational_quadratic_spline.py:180-288)."""
        if self.first:
            x, log_det = mf.real_line_to_interval(x, log_det, self.lo, self.hi)
        y, lad, b = self._spline(x, params, False)
        return y, log_det + lad, b


# =====================================================================================  sphere wrapper
class _SphereLayer:
    """sphere_base.inv_flow_mapping / flow_mapping (sphere_base.py:601-695), householder rotation mode."""
    dim = 1

    def _setup_base(self, o, first, embedding, n_hh_iter=-1):
        self.first = first
        self.embedding = embedding          # always_parametrize_in_embedding_space
        self.add_rotation = o.get("add_rotation", 0)
        self.n_hh_iter = 0
        self.n_rot = 0
        self.rotation_mode = o.get("rotation_mode", "householder")
        if self.add_rotation:
            E = self.dim + 1
            if self.rotation_mode == "householder":
                self.n_hh_iter = E if n_hh_iter == -1 else n_hh_iter
                self.n_rot = self.n_hh_iter * E
            else:
                self.n_rot = {"angles": E * (E - 1) // 2, "xyz": 3, "quaternion": 4}[self.rotation_mode]

    def _rotate(self, x, log_det, params, transpose):
        if not self.embedding:
            x, log_det = mf.spherical_to_eucl(x, log_det, self.dim)
        R = mf.rotation_matrix(params[:, :self.n_rot], self.dim, self.n_hh_iter, self.rotation_mode)
        x = matvec(R, x, transpose=transpose)
        if not self.embedding:
            x, log_det = mf.eucl_to_spherical(x, log_det, self.dim)
        return x, log_det

    def inverse(self, x, log_det, params, fix_first=None):
        if self.add_rotation:
            x, log_det = self._rotate(x, log_det, params, True)
        self._cur_rot = params[:, :self.n_rot]        # 'f' may read kappa off the rotation parameters (fvm_2d.py:289-330)
        x, log_det, bins = self._core_inverse(x, log_det, params[:, self.n_rot:])
        first = self.first if fix_first is None else fix_first
        if first:
            if self.embedding:
                x, log_det = mf.eucl_to_spherical(x, log_det, self.dim)
            x, log_det = mf.sphere_to_plane(x, log_det, self.dim)
        return x, log_det, bins

    def forward(self, x, log_det, params, fix_first=False):
        first = fix_first if fix_first else self.first
        if first:
            x, log_det = mf.plane_to_sphere(x, log_det, self.dim)
            if self.embedding:
                x, log_det = mf.spherical_to_eucl(x, log_det, self.dim)
        self._cur_rot = params[:, :self.n_rot]
        x, log_det, bins = self._core_forward(x, log_det, params[:, self.n_rot:])
        if self.add_rotation:
            x, log_det = self._rotate(x, log_det, params, False)
        return x, log_det, bins

    def embed(self, x):
        """_embedding_conditional_return (sphere_base.py:786-791)."""
        if x.shape[1] == self.dim:
            x, _ = mf.spherical_to_eucl(x, 0.0, self.dim)
        return x

    def rot_row_from_state(self, sd, prefix):
        return sd[prefix + "householder_params"].reshape(-1) if self.n_rot else np.zeros(0)


# =====================================================================================  'o'
class OLayer(_SphereLayer, _SplineParamMixin):
    dim = 1

    def __init__(self, dimension, o, first, embedding):
        assert dimension == 1
        self._setup_base(o, first, embedding)
        self._setup_counts(o, circular=True)
        self.natural_direction = o["natural_direction"]
        self.total_param_num = self.n_rot + self.n_spline

    def row_from_state(self, sd, prefix):
        return np.concatenate([self.rot_row_from_state(sd, prefix), self._row_from_state(sd, prefix)])[None, :]

    def _spline(self, x, params, use_inverse):
        w, h, d = self._unpack_whd(params)
        if self.smooth == 0:
            if self.fix_bd > 0.0:
                fl = np.ones(w.shape[:-1] + (1,)) * self.bd_fixed_value
                d = np.concatenate([fl, fl] if d is None else [fl, d, fl], axis=-1)
            else:
                d = np.concatenate([d, d[:, 0:1]], axis=-1)
            return splines.rqs_plain(x, w, h, d, use_inverse, 0.0, TWO_PI, 0.0, TWO_PI, self.min_w, self.min_h, self.min_d)
        return splines.rqs_circular(x, w, h, use_inverse, self.min_w, self.min_h)

    def _core_inverse(self, x, log_det, params):
        """splines_1d.py:111-207 (returns log_det_new: the embedding conversions' log-det is dropped, and is 0 on S1)."""
        if self.embedding:
            x, log_det = mf.eucl_to_spherical(x, log_det, 1)
        x = mf.safe_angle_within_2pi(x)
        y, lad, b = self._spline(x, params, self.natural_direction != 0)
        log_det = log_det + lad.sum(axis=-1)
        y = mf.safe_angle_within_2pi(y)
        if self.embedding:
            y, _ = mf.spherical_to_eucl(y, log_det, 1)
        return y, log_det, [b]

    def _core_forward(self, x, log_det, params):
        """splines_1d.py:210-306."""
        if self.embedding:
            x, log_det = mf.eucl_to_spherical(x, log_det, 1)
        x = np.clip(x, 0.0, TWO_PI)
        y, lad, b = self._spline(x, params, self.natural_direction == 0)
        log_det = log_det + lad.sum(axis=-1)
        y = np.clip(y, 0.0, TWO_PI)
        if self.embedding:
            y, _ = mf.spherical_to_eucl(y, log_det, 1)
        return y, log_det, [b]


# =====================================================================================  'm'
MIN_OMEGA, MAX_OMEGA = 0.001, 0.999


def _moebius_omega(pars):
    """omega vector / length from (B,nc,4) = (wx, wy, logit-length, log-weight), or (B,nc,3) = (omega angle, logit-length, log-weight) for
    use_moebius_xyz_parametrization=False  (moebius_1d.py:157-178)."""
    loglen = pars[:, :, -2:-1]
    denom = np.logaddexp(0.0, -loglen)
    length = MIN_OMEGA + np.exp(np.log(MAX_OMEGA - MIN_OMEGA) - denom)
    if pars.shape[2] == 4:
        vec = pars[:, :, :2] / np.sqrt((pars[:, :, :2] ** 2).sum(axis=2, keepdims=True)) * length
    else:
        vec = np.concatenate([np.cos(pars[:, :, 0:1]) * length, np.sin(pars[:, :, 0:1]) * length], axis=2)
    return vec, length


def moebius_trafo(x, pars):
    """simple_moebius_trafo (moebius_1d.py:140-216).  x (B,1) in [-pi,pi]."""
    cx, sx = np.cos(x)[:, None, :], np.sin(x)[:, None, :]
    cmp_, smp = np.cos(-PI), np.sin(-PI)
    vec, length = _moebius_omega(pars)
    omo = 1.0 - length ** 2
    opo = 1.0 + length ** 2 - 2 * (cx * vec[:, :, 0:1] + sx * vec[:, :, 1:2])
    opo_mp = 1.0 + length ** 2 - 2 * (cmp_ * vec[:, :, 0:1] + smp * vec[:, :, 1:2])
    y_mp = omo * (smp - vec[:, :, 1:2]) - vec[:, :, 1:2] * opo_mp
    x_mp = omo * (cmp_ - vec[:, :, 0:1]) - vec[:, :, 0:1] * opo_mp
    rot = -PI - np.arctan2(y_mp, x_mp)
    yv = omo * (sx - vec[:, :, 1:2]) - vec[:, :, 1:2] * opo
    xv = omo * (cx - vec[:, :, 0:1]) - vec[:, :, 0:1] * opo
    xp = np.cos(rot) * xv - np.sin(rot) * yv
    yp = np.sin(rot) * xv + np.cos(rot) * yv
    arc = np.arctan2(yp, xp)[:, :, -1:] + PI
    ln = pars[:, :, -1:]
    weighted = arc * np.exp(ln - logsumexp(ln, axis=1, keepdims=True))
    return weighted.sum(axis=1) - PI


def moebius_deriv(x, pars):
    """simple_moebius_trafo_deriv (moebius_1d.py:219-259)."""
    cx, sx = np.cos(x)[:, None, :], np.sin(x)[:, None, :]
    vec, length = _moebius_omega(pars)
    omo = 1.0 - length ** 2
    opo = 1.0 + length ** 2 - 2 * (cx * vec[:, :, 0:1] + sx * vec[:, :, 1:2])
    ln = pars[:, :, -1:]
    wd = (np.log(omo / opo) + ln) - logsumexp(ln, axis=1, keepdims=True)
    return np.exp(logsumexp(wd, axis=1))


def moebius_bisection_newton(z, pars, n_bisect=20, n_newton=20, tol=1e-14):
    """inverse_bisection_n_newton (bisection_n_newton.py:137-256) with [-pi, pi] bounds (moebius_1d.py:78,125)."""
    hi = np.full_like(z, PI)
    lo = np.full_like(z, -PI)
    mid = None
    for _ in range(n_bisect):
        mid = (hi + lo) / 2.0
        f = moebius_trafo(mid, pars)
        right = (f < z).astype(z.dtype)
        leftp = 1.0 - right
        ok = (np.abs(f - z) <= 1e-6 * np.abs(z)).astype(z.dtype)
        lo = (1.0 - ok) * (right * mid + leftp * lo) + ok * mid
        hi = (1.0 - ok) * (right * hi + leftp * mid) + ok * mid
    prev = mid.copy()
    active = np.ones(z.shape[0], dtype=bool)
    for _ in range(n_newton):
        pp = pars[active] if pars.shape[0] > 1 else pars
        f_eval = moebius_trafo(prev[active], pp) - z[active]
        upd = f_eval / moebius_deriv(prev[active], pp)
        prev[active] = prev[active] - upd
        idx = np.nonzero(active)[0]
        active[idx] = np.abs(upd).sum(axis=1) >= tol
        if not active.any():
            break
    return prev


class MLayer(_SphereLayer):
    dim = 1

    def __init__(self, dimension, o, first, embedding):
        assert dimension == 1
        self._setup_base(o, first, embedding)
        self.nc = o["num_basis_functions"]
        self.natural_direction = o["natural_direction"]
        self.n_omega = 4 if o.get("use_moebius_xyz_parametrization", 1) else 3
        self.total_param_num = self.n_rot + self.n_omega * self.nc

    def row_from_state(self, sd, prefix):
        return np.concatenate([self.rot_row_from_state(sd, prefix), sd[prefix + "moebius_pars"].reshape(-1)])[None, :]

    def _run(self, x, log_det, params, do_direct):
        pars = params.reshape(params.shape[0], self.nc, self.n_omega)
        if self.embedding:
            x, log_det = mf.eucl_to_spherical(x, log_det, 1)
        x = np.where(x > PI, x - TWO_PI, x)                        # 0..2pi -> -pi..pi   (moebius_1d.py:73-74)
        if do_direct:
            ld = np.log(moebius_deriv(x, pars)).sum(axis=-1)
            x = moebius_trafo(x, pars)
        else:
            x = moebius_bisection_newton(x, pars)
            ld = -np.log(moebius_deriv(x, pars)).sum(axis=-1)
        x = np.where(x < 0, TWO_PI + x, x)
        log_det = log_det + ld
        if self.embedding:
            x, log_det = mf.spherical_to_eucl(x, log_det, 1)
        return x, log_det, []

    def _core_inverse(self, x, log_det, params):
        """moebius_1d.py:57-99."""
        return self._run(x, log_det, params, do_direct=not self.natural_direction)

    def _core_forward(self, x, log_det, params):
        """moebius_1d.py:101-138."""
        return self._run(x, log_det, params, do_direct=bool(self.natural_direction))
