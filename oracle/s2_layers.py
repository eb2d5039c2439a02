"""2-sphere layers 'f' (layers/spheres/fvm_2d.py) and 'v' (layers/spheres/exponential_map_s2.py) restated in numpy.
Oracle = test infrastructure only."""
import numpy as np

from . import manifolds as mf
from . import splines
from .sphere_layers import _SphereLayer
from .special import bounded_log_fn, logsumexp

PI = np.pi


def azimuthal_scaling(c):
    """fvm_2d.py:267-271  smooth step that switches the circular flow off at the poles."""
    return np.where(c <= 0, 6 * c ** 5 + 15 * c ** 4 + 10 * c ** 3 + 1.0, -6 * c ** 5 + 15 * c ** 4 - 10 * c ** 3 + 1.0)


class FLayer(_SphereLayer):
    dim = 2

    def __init__(self, dimension, o, first, embedding, nested_factory):
        assert dimension == 2
        self._setup_base(o, first, embedding, n_hh_iter=o["num_householder_iter"])
        self.extra_rotation = bool(o["add_extra_rotation_inbetween"])
        self.kappa_prediction = o["kappa_prediction"]
        self.kappa_clamping = o["kappa_clamping"]
        self.own_kappa = self.kappa_prediction in ("direct_log_real_bounded", "softplus_real_bounded", "log_bounded")
        self.zs = -1.0 if o["inverse_z_scaling"] else 1.0
        self.min_kappa = o["min_kappa"]
        self.region = o["boundary_cos_theta_identity_region"]
        self.vertical = self.circular = self.correlated = None
        n = 1 if self.own_kappa else 0
        lo, hi = -(1.0 - self.region), (1.0 - self.region)
        ival = "i1_-%.2f_%.2f" % (1.0 - self.region, 1.0 - self.region)
        if o["add_vertical_rq_spline_flow"]:
            fd = {"r": {"fix_boundary_derivatives": -1.0 if o["vertical_fix_boundary_derivative"] == 0 else 1.0,
                        "smooth_second_derivative": o["vertical_smooth"],
                        "restrict_max_min_width_height_ratio": o["vertical_restrict_max_min_width_height_ratio"],
                        "fix_first_width_n_height_to_zero": o["vertical_fix_first_width_n_height_to_zero"],
                        "also_fix_second_width_to_zero": o["vertical_also_fix_second_width_to_zero"],
                        "independent_width_height_parametrization": o["vertical_independent_width_height_parametrization"]}}
            if o["spline_num_basis_functions"] == -1:
                for i in range(len(o["vertical_flow_defs"])):
                    fd[(0, i)] = {"r": dict(fd["r"], num_basis_functions=3 if i % 2 == 1 else 2)}
            else:
                fd["r"]["num_basis_functions"] = o["spline_num_basis_functions"]
            self.vertical = nested_factory(ival, o["vertical_flow_defs"], fd)        # fvm_2d.py:193-197
            n += self.vertical.total_number_amortizable_params
        self.n_vertical = n - (1 if self.own_kappa else 0)
        if o["add_circular_rq_spline_flow"]:
            assert o["circular_add_rotation"] == 0
            fd = {"o": {"num_basis_functions": 2, "smooth_second_derivative": 1,
                        "fix_first_width_n_height_to_zero": o["vertical_fix_first_width_n_height_to_zero"],
                        "also_fix_second_width_to_zero": o["vertical_also_fix_second_width_to_zero"],
                        "independent_width_height_parametrization": o["vertical_independent_width_height_parametrization"],
                        "add_rotation": 0}}
            self.circular = nested_factory("s1", o["circular_flow_defs"], fd)        # fvm_2d.py:221-225
            self.n_circular = self.circular.total_number_amortizable_params
            n += self.n_circular
        if o["add_correlated_rq_spline_flow"]:
            assert self.vertical is None and self.circular is None
            self.correlated = nested_factory(ival + "+s1", o["vertical_flow_defs"] + "+" + o["circular_flow_defs"], {},
                                             mlp_dims="64", mlp_ranks=o["correlated_max_rank"])   # fvm_2d.py:250-256
            n += self.correlated.total_number_amortizable_params
        self.total_param_num = self.n_rot + n

    def row_from_state(self, sd, prefix):
        parts = [self.rot_row_from_state(sd, prefix)] + ([sd[prefix + "loglike_kappa"].reshape(-1)] if self.own_kappa else [])
        if self.correlated is not None:
            parts.append(sd[prefix + "correlated_flow_params"].reshape(-1))
        if self.vertical is not None:
            parts.append(sd[prefix + "vertical_flow_params"].reshape(-1))
        if self.circular is not None:
            parts.append(sd[prefix + "circular_flow_params"].reshape(-1))
        return np.concatenate(parts)[None, :]

    def _split(self, params):
        # kappa_fn / kappa from the rotation parameters (fvm_2d.py:105-139, 289-330)
        from .special import softplus
        if self.own_kappa:
            p0 = params[:, 0:1]
            if self.kappa_prediction == "direct_log_real_bounded":
                kappa = np.exp(np.maximum(p0, -5.0) if self.kappa_clamping else p0) + self.min_kappa
            elif self.kappa_prediction == "softplus_real_bounded":
                kappa = softplus(np.maximum(p0, -5.0) if self.kappa_clamping else p0) + self.min_kappa
            else:
                v = softplus(p0)
                kappa = np.exp((np.maximum(v, -5.0) if self.kappa_clamping else v) + np.log(self.min_kappa))
            rest = params[:, 1:]
        else:
            rot = self._cur_rot
            sq = (rot ** 2).sum(axis=1, keepdims=True) if self.kappa_prediction.startswith("mu") else (rot[:, 1:] ** 2).sum(axis=1, keepdims=True)
            kappa = sq if self.kappa_prediction.endswith("squared") else np.sqrt(sq)
            rest = params
        vert = rest[:, :self.n_vertical] if self.vertical is not None else None
        circ = rest[:, self.n_vertical:self.n_vertical + self.n_circular] if self.circular is not None else None
        corr = rest if self.correlated is not None else None
        return kappa, vert, circ, corr

    def _masked(self, flow, direction, x, log_det, pars, mask):
        """evaluate a nested passthrough flow on the rows in `mask` only (identity region, fvm_2d.py:436-482)."""
        if mask is None:
            return getattr(flow, direction)(x, log_det, None, pars)
        if mask.sum() == 0:
            return x, log_det, []
        p = pars[mask] if pars.shape[0] > 1 else pars
        xs, lds, b = getattr(flow, direction)(x[mask], log_det[mask], None, p)
        x, log_det = x.copy(), log_det.copy()
        x[mask] = xs
        log_det[mask] = lds
        return x, log_det, b

    _INBETWEEN = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]])        # fvm_2d.py:392, 679

    def _inbetween(self, cos_theta, angle, log_det, inverse):
        """add_extra_rotation_inbetween (fvm_2d.py:381-402 inverse with the transposed matrix, :664-688 forward): the pole moves onto the
        equator between the kappa step and the nested spline flows"""
        th = np.arccos(cos_theta)
        log_det = log_det - np.log(np.sin(mf.safe_angle_within_pi(th[:, 0])))
        comb, log_det = mf.spherical_to_eucl(np.concatenate([th, angle], axis=1), log_det, 2)
        m = self._INBETWEEN.T if inverse else self._INBETWEEN
        comb = comb @ m.T
        comb, log_det = mf.eucl_to_spherical(comb, log_det, 2)
        log_det = log_det + np.log(np.sin(mf.safe_angle_within_pi(comb[:, 0])))
        return np.cos(comb[:, :1]), comb[:, 1:], log_det

    def _core_inverse(self, x, log_det, params):
        """fisher_von_mises_2d._inv_flow_mapping (fvm_2d.py:273-500)."""
        bins = []
        if self.embedding:
            x, log_det = mf.eucl_to_spherical(x, log_det, 2)
        kappa, vert, circ, corr = self._split(params)
        prev = np.cos(x[:, :1])
        log_det = log_det + np.log(np.sin(mf.safe_angle_within_pi(x[:, 0])))
        small = kappa < 100
        with np.errstate(over="ignore"):
            safe = np.where(small, np.log(np.exp(2 * np.where(small, kappa, 1.0)) - 1.0), 2 * kappa)
        ld_upd = (np.log(2 * kappa) + kappa * (self.zs * prev + 1) - safe)[:, 0]
        ret = self.zs * ((1.0 + np.exp(-2 * kappa) - 2 * np.exp(kappa * (self.zs * prev - 1))) / (-1 + np.exp(-2 * kappa)))
        ret = np.where(kappa < 1e-8, prev, ret)
        log_det = log_det + ld_upd
        ret = mf.safe_costheta(ret)
        angle = x[:, 1:]
        if self.extra_rotation:
            ret, angle, log_det = self._inbetween(ret, angle, log_det, True)
        mask = None
        if self.region != 0.0:
            mask = ((ret > (-1.0 + self.region)) & (ret < (1.0 - self.region)))[:, 0]
        if corr is not None:
            comb = np.concatenate([ret, angle], axis=1)
            comb, log_det, b = self._masked(self.correlated, "all_layer_inverse", comb, log_det, corr, mask)
            bins += b
            ret, angle = comb[:, :1], comb[:, 1:]
        else:
            if self.circular is not None:
                sc = azimuthal_scaling(ret)
                cp = np.broadcast_to(circ, (ret.shape[0], circ.shape[1])) * sc     # all 'o' params scale: no rotation params
                angle, log_det, b = self._masked(self.circular, "all_layer_inverse", angle, log_det, cp, mask)
                bins += b
            if vert is not None:
                ret, log_det, b = self._masked(self.vertical, "all_layer_inverse", ret, log_det, vert, mask)
                bins += b
        ret = mf.safe_costheta(ret)
        ret = np.arccos(ret)
        log_det = log_det - np.log(np.sin(mf.safe_angle_within_pi(ret[:, 0])))
        out = np.concatenate([ret, angle], axis=1)
        if self.embedding:
            out, log_det = mf.spherical_to_eucl(out, log_det, 2)
        return out, log_det, bins

    def _core_forward(self, x, log_det, params):
        """fisher_von_mises_2d._flow_mapping (fvm_2d.py:502-726)."""
        bins = []
        if self.embedding:
            x, log_det = mf.eucl_to_spherical(x, log_det, 2)
        kappa, vert, circ, corr = self._split(params)
        prev = np.cos(x[:, :1])
        log_det = log_det + np.log(np.sin(mf.safe_angle_within_pi(x[:, 0])))
        angle = x[:, 1:]
        mask = None
        if self.region != 0.0:
            mask = ((prev > (-1.0 + self.region)) & (prev < (1.0 - self.region)))[:, 0]
        if corr is not None:
            comb = np.concatenate([prev, angle], axis=1)
            comb, log_det, b = self._masked(self.correlated, "all_layer_forward", comb, log_det, corr, mask)
            bins += b
            prev, angle = comb[:, :1], comb[:, 1:]
        else:
            if vert is not None:
                prev, log_det, b = self._masked(self.vertical, "all_layer_forward", prev, log_det, vert, mask)
                bins += b
            if self.circular is not None:
                sc = azimuthal_scaling(prev)
                cp = np.broadcast_to(circ, (prev.shape[0], circ.shape[1])) * sc
                angle, log_det, b = self._masked(self.circular, "all_layer_forward", angle, log_det, cp, mask)
                bins += b
        if self.extra_rotation:
            prev, angle, log_det = self._inbetween(prev, angle, log_det, False)
        log_det = log_det - np.log(kappa * self.zs * prev + kappa / np.tanh(kappa))[:, 0]
        ret = self.zs * (1.0 + (1.0 / kappa) * np.log(0.5 * (1.0 + self.zs * prev) + (0.5 - 0.5 * self.zs * prev) * np.exp(-2.0 * kappa)))
        ret = np.where(kappa < 1e-8, prev, ret)
        ret = mf.safe_costheta(ret)
        ret = np.arccos(ret)
        log_det = log_det - np.log(np.sin(mf.safe_angle_within_pi(ret[:, 0])))
        out = np.concatenate([ret, angle], axis=1)
        if self.embedding:
            out, log_det = mf.spherical_to_eucl(out, log_det, 2)
        return out, log_det, bins


# =====================================================================================  'v'
def _mu_norm_fn(x, max_value=1.0, stretch=10.0):
    """generate_normalization_function (exponential_map_s2.py:32-43)."""
    return -np.log(1.0 + (np.e - 1.0) * np.exp(-x / stretch)) + max_value


class VLayer(_SphereLayer):
    dim = 2

    def __init__(self, dimension, o, first, embedding):
        assert dimension == 2
        self._setup_base(o, first, embedding)
        if o["mean_parametrization"] != "old":
            raise NotImplementedError("oracle: v mean_parametrization")
        self.kind = o["exp_map_type"]
        if self.kind not in ("exponential", "linear", "quadratic", "splines"):
            raise NotImplementedError("oracle: v exp_map_type %s" % self.kind)
        self.nc = o["num_components"]
        self.natural_direction = o["natural_direction"]
        self.max_newton = o["max_num_newton_iter"]
        self.npp = 3 + {"exponential": 2, "splines": 1 + 3 * 10 + 1}.get(self.kind, 1)      # exponential_map_s2.py:124-129
        self.total_param_num = self.n_rot + self.npp * self.nc

    def row_from_state(self, sd, prefix):
        return np.concatenate([self.rot_row_from_state(sd, prefix), sd[prefix + "potential_pars"].reshape(-1)])[None, :]

    # ---- exponential map + Jacobian (exponential_map_s2.py:248-442)
    def exp_map(self, x, pp):
        norm = np.sqrt((pp[:, :3, :] ** 2).sum(axis=1, keepdims=True))
        mu = pp[:, :3, :] / norm
        fake = _mu_norm_fn(norm)
        lw = pp[:, 3:4, :] - logsumexp(pp[:, 3:4, :], axis=2, keepdims=True) + np.log(fake)
        w = np.exp(lw)
        xmu = (x[:, :, None] * mu).sum(axis=1, keepdims=True)              # (B,1,nc)
        if self.kind == "exponential":
            beta = np.exp(pp[:, 4:5, :])
            e = np.exp(beta * (xmu - 1.0))
            grad = (w * mu * e).sum(axis=-1)
            gj = np.einsum("biu,bju->bij", beta * w * mu * e, np.broadcast_to(mu, (x.shape[0],) + mu.shape[1:]))
        elif self.kind == "splines":                                             # (:346-388): d potential / d (mu . x) is an RQ spline on [-1, 1]
            t = lambda a: np.transpose(a, (0, 2, 1))
            res, lad, _ = splines.rqs_plain(t(xmu), t(pp[:, 4:14, :]), t(pp[:, 14:24, :]), t(pp[:, 24:35, :]), False, -1.0, 1.0, -1.0, 1.0,
                                            min_w=1e-3, min_h=1e-3, min_d=1e-3)
            res, deriv = t(res), t(np.exp(lad))
            grad = (w * mu * res).sum(axis=-1)
            gj = np.einsum("biu,bju->bij", w * mu * deriv, np.broadcast_to(mu, (x.shape[0],) + mu.shape[1:]))
        elif self.kind == "linear":
            grad = np.broadcast_to((w * mu).sum(axis=-1), x.shape)
            gj = None
        else:
            grad = (w * mu * xmu).sum(axis=-1)
            gj = np.einsum("biu,bju->bij", np.broadcast_to(w * mu, (x.shape[0],) + mu.shape[1:]),
                           np.broadcast_to(mu, (x.shape[0],) + mu.shape[1:]))
        # unnormalized logarithmic map with Jacobians (exponential_map_s2.py:163-219)
        tn = np.sqrt((grad ** 2).sum(axis=1, keepdims=True))
        nt = grad / tn
        ca = (nt * x).sum(axis=1, keepdims=True)
        alpha = np.arccos(ca)
        sa = np.sin(alpha)
        tv = (nt - x * ca) / sa
        proj = (grad * tv).sum(axis=1, keepdims=True)
        eye = np.eye(3)[None]
        d_t_d_base = eye * (-ca / sa)[:, :, None]
        d_t_d_theta = ((x - nt * ca) / (sa ** 2))[:, :, None]
        inv_sq = -1.0 / np.sqrt(1.0 - ca ** 2)
        d_theta_d_base = (inv_sq * nt)[:, None, :]
        jac_t = d_t_d_base + d_t_d_theta @ d_theta_d_base
        jac_p = (jac_t * grad[:, :, None]).sum(axis=1, keepdims=True)
        if gj is not None:
            d_theta_d_norm = (inv_sq * x)[:, None, :]
            d_norm_d_un = (-grad / tn ** 2)[:, :, None] @ nt[:, None, :] + eye * (1.0 / tn)[:, :, None]
            d_t_d_norm = eye * (1.0 / sa)[:, :, None]
            jac_t = jac_t + d_t_d_theta @ d_theta_d_norm @ d_norm_d_un @ gj
            jac_t = jac_t + d_t_d_norm @ d_norm_d_un @ gj
            jac_p = jac_p + (tv[:, :, None] * gj).sum(axis=1, keepdims=True)
        res = x * np.cos(proj) + tv * np.sin(proj)
        outer = (-x * np.sin(proj))[:, :, None] @ jac_p
        first = eye * np.cos(proj)[:, :, None] + outer
        second = jac_t * np.sin(proj)[:, :, None] + (tv * np.cos(proj))[:, :, None] @ jac_p
        jac = first + second
        t2 = np.cross(x, tv, axis=1)
        basis = np.stack([tv, t2], axis=2)
        pj = jac @ basis
        return res, np.transpose(pj, (0, 2, 1)) @ pj, jac, tv

    @staticmethod
    def _log_map(base, target):
        """basic_logarithmic_map (exponential_map_s2.py:221-244)."""
        alt = np.zeros_like(base)
        alt[:, 0] = 1.0
        ca = (target * base).sum(axis=1, keepdims=True)
        conv = ca >= 1
        ca = np.where(conv, (target * alt).sum(axis=1, keepdims=True), ca)
        alpha = np.arccos(ca)
        used = np.where(conv, alt, base)
        tv = (target - used * ca) / np.sin(alpha)
        return tv, np.where(conv, 0.0, alpha)

    def _newton_fast(self, target, pp):
        """inverse_bisection_n_newton_sphere_fast (bisection_n_newton.py:394-465): damping 0.4, row masking."""
        prev = np.zeros_like(target)
        prev[:, 2] = -1.0
        active = np.ones(target.shape[0], dtype=bool)
        for _ in range(self.max_newton):
            p = pp[active] if pp.shape[0] > 1 else pp
            phi, _, jac, _ = self.exp_map(prev[active], p)
            tg = target[active]
            fn = -(phi * tg).sum(axis=-1, keepdims=True) + 1.0
            rv = -(np.transpose(jac, (0, 2, 1)) @ tg[:, :, None])[:, :, 0]
            gn = np.sqrt((rv ** 2).sum(axis=1, keepdims=True))
            nv, alpha = self._log_map(prev[active], -(rv / gn))
            gp = (nv * rv).sum(axis=1, keepdims=True)
            proj = -(fn / gp)
            proj = np.where(alpha == 0, 0.0, proj)
            prev[active] = prev[active] * np.cos(0.4 * proj) + nv * np.sin(0.4 * proj)
            idx = np.nonzero(active)[0]
            active[idx] = np.abs(proj[:, 0]) >= 1e-12
            if not active.any():
                break
        return prev

    def _newton_slow(self, target, pp):
        """inverse_bisection_n_newton_sphere (bisection_n_newton.py:330-391): damping 0.1, global early exit."""
        prev = np.zeros_like(target)
        prev[:, 2] = -1.0
        for _ in range(self.max_newton):
            phi, _, jac, _ = self.exp_map(prev, pp)
            fn = -(phi * target).sum(axis=-1, keepdims=True) + 1.0
            rv = -(np.transpose(jac, (0, 2, 1)) @ target[:, :, None])[:, :, 0]
            gn = np.sqrt((rv ** 2).sum(axis=1, keepdims=True))
            nv, alpha = self._log_map(prev, -(rv / gn))
            gp = (nv * rv).sum(axis=1, keepdims=True)
            proj = -(fn / gp)
            if proj.max() < 1e-12:
                break
            prev = prev * np.cos(0.1 * proj) + nv * np.sin(0.1 * proj)
        return prev

    def _pp(self, params):
        return params.reshape(params.shape[0], self.npp, self.nc)

    def _core_inverse(self, x, log_det, params):
        """exponential_map_s2._inv_flow_mapping (exponential_map_s2.py:446-487)."""
        pp = self._pp(params)
        if not self.embedding:
            x, log_det = mf.spherical_to_eucl(x, log_det, 2)
        if self.natural_direction:
            res = self._newton_slow(x, pp)
            _, j2, _, _ = self.exp_map(res, pp)
            log_det = log_det - 0.5 * np.linalg.slogdet(j2)[1]
        else:
            res, j2, _, _ = self.exp_map(x, pp)
            log_det = log_det + 0.5 * np.linalg.slogdet(j2)[1]
        if not self.embedding:
            res, log_det = mf.eucl_to_spherical(res, log_det, 2)
        return res, log_det, []

    def _core_forward(self, x, log_det, params):
        """exponential_map_s2._flow_mapping (exponential_map_s2.py:489-528)."""
        pp = self._pp(params)
        if not self.embedding:
            x, log_det = mf.spherical_to_eucl(x, log_det, 2)
        if self.natural_direction:
            res, j2, _, _ = self.exp_map(x, pp)
            log_det = log_det + 0.5 * np.linalg.slogdet(j2)[1]
        else:
            res = self._newton_fast(x, pp)
            _, j2, _, _ = self.exp_map(res, pp)
            log_det = log_det - 0.5 * np.linalg.slogdet(j2)[1]
        if not self.embedding:
            res, log_det = mf.eucl_to_spherical(res, log_det, 2)
        return res, log_det, []
