"""Gaussianization-flow layer 'g' (layers/euclidean/gaussianization_flow.py + euclidean_base.py +
bisection_n_newton.py:11-135) restated in numpy.  Oracle = test infrastructure only.

Parameter row layout (SURVEY 8a'):  [offset D if model_offset][rotation][means (K-center_mean)*D][log_widths K*D][log_weights K*D]
[log skew exponents K*D if add_skewness]  (K-major (K,D)); rq_splines variant: [offset][rotation][log_w D*K][log_h D*K][log_d D*(K+1)][box D*4].
Rotation section: householder hh*D | angles D(D-1)/2 | cayley 1 | triangular_combination [lower D(D-1)/2][diag D-1][upper D(D-1)/2].
"""
import itertools

import numpy as np

from . import splines
from .special import bounded_log_fn, erfinv, householder_matrix, logsumexp, matvec, softplus

PADE_BOUND = 0.5e-7   # gaussianization_flow.py:140
PADE_A = 0.147        # gaussianization_flow.py:143


class GfSpec:
    """Option set of one 'g' layer (names/defaults: flow_options.py:32-54; constructor gaussianization_flow.py:51-70)."""

    def __init__(self, dimension, opts, model_offset):
        self.D = dimension
        self.K = opts["num_kde"]
        self.rotation_mode = opts["rotation_mode"]
        hh = opts["num_householder_iter"]
        self.hh_iter = dimension if hh == -1 else hh
        if self.rotation_mode == "householder":
            self.n_rot = self.hh_iter * dimension
        elif self.rotation_mode == "none" or dimension < 2:
            self.n_rot = 0
        elif self.rotation_mode == "angles":                                     # gaussianization_flow.py:198-209
            self.n_rot = dimension * (dimension - 1) // 2
        elif self.rotation_mode == "cayley":                                     # :211-223
            assert dimension == 2
            self.n_rot = 1
        elif self.rotation_mode == "triangular_combination":                     # :158-170
            self.n_rot = dimension - 1 + dimension * (dimension - 1)
        else:
            raise NotImplementedError("oracle: rotation_mode %s" % self.rotation_mode)
        self.fit_normalization = opts["fit_normalization"]
        self.regulate_normalization = opts["regulate_normalization"]
        self.inverse_function_type = opts["inverse_function_type"]
        self.model_offset = model_offset
        self.softplus_for_width = opts["softplus_for_width"]
        self.width_smooth_saturation = opts["width_smooth_saturation"]
        self.clamp_widths = opts["clamp_widths"]
        self.width_min = opts["lower_bound_for_widths"]
        self.width_max = opts["upper_bound_for_widths"] if opts["upper_bound_for_widths"] > 0 else None
        self.norm_min = opts["lower_bound_for_norms"]
        self.norm_max = opts["upper_bound_for_norms"]
        self.stretch = opts["nonlinear_stretch_type"]
        self.center_mean = opts["center_mean"]
        self.add_skewness = opts["add_skewness"]
        kd = self.K * self.D
        if self.stretch == "classic":
            n = 2 * kd + (kd if self.fit_normalization else 0) + (kd if self.add_skewness else 0) - (self.D if self.center_mean else 0)
        else:
            n = 2 * kd + (self.K + 1) * self.D + 4 * self.D
        self.total_param_num = (self.D if model_offset else 0) + self.n_rot + n

    # state_dict -> flat row in extra_inputs layout (gaussianization_flow.py:1169-1222 in reverse)
    def row_from_state(self, sd, prefix):
        parts = []
        if self.model_offset:
            parts.append(np.asarray(sd[prefix + "offsets"], dtype=np.float64).reshape(-1))
        if self.n_rot:
            name = {"householder": "vs", "angles": "angle_pars", "cayley": "cayley_pars", "triangular_combination": "triangle_trafo_pars"}
            parts.append(sd[prefix + name[self.rotation_mode]].reshape(-1))
        if self.stretch == "classic":
            parts.append(sd[prefix + "kde_means"].reshape(-1))
            parts.append(sd[prefix + "kde_log_widths"].reshape(-1))
            if self.fit_normalization:
                parts.append(sd[prefix + "kde_log_weights"].reshape(-1))
            if self.add_skewness:
                parts.append(sd[prefix + "kde_log_skew_exponents"].reshape(-1))
        else:
            for k in ("log_widths", "log_heights", "log_derivatives", "boundary_points"):
                parts.append(sd[prefix + k].reshape(-1))
        row = np.concatenate([np.asarray(p, dtype=np.float64) for p in parts])[None, :]
        assert row.shape[1] == self.total_param_num
        return row


def _width_regulator(spec, x):
    """gaussianization_flow.py:269-317."""
    lo_clamp = np.log(0.01 * spec.width_min)
    if spec.softplus_for_width:
        if spec.clamp_widths:
            hi = np.log(spec.width_max) if spec.width_max is not None else None
            x = np.clip(x, lo_clamp, hi)
        return np.log(softplus(x) + spec.width_min)
    if spec.width_smooth_saturation == 0:
        if spec.clamp_widths:
            hi = np.log(spec.width_max) if spec.width_max is not None else None
            x = np.clip(x, lo_clamp, hi)
        return np.log(np.exp(x) + spec.width_min)
    if spec.clamp_widths:
        x = np.clip(x, lo_clamp, np.log(spec.width_max) * 3.0)
    return bounded_log_fn(x, spec.width_min, spec.width_max, center=True)


def _unit_lower(D, entries):
    """matrix_fns.obtain_lower_triangular_matrix_and_logdet (layers/matrix_fns.py:27-55) with a unit diagonal: the entries fill the
    sub-diagonals from the bottom-left corner upwards."""
    m = np.broadcast_to(np.eye(D), (entries.shape[0], D, D)).copy()
    c = 0
    for ind in range(D - 1):
        off = D - 1 - ind
        for j in range(ind + 1):
            m[:, j + off, j] = entries[:, c + j]
        c += ind + 1
    return m


def rotation(spec, rp):
    """the rotation section of the row -> ("matrix", Q) or ("triangular", (L, diag, U))  (gaussianization_flow.py:711-799)."""
    D = spec.D
    if spec.rotation_mode == "householder":
        return "matrix", householder_matrix(rp.reshape(-1, spec.hh_iter, D))
    if spec.rotation_mode == "angles":                                           # Givens rotations, prev = new @ prev (:760-780)
        q = np.broadcast_to(np.eye(D), (rp.shape[0], D, D)).copy()
        for ind, (a, b) in enumerate(itertools.combinations(range(D), 2)):
            g = np.broadcast_to(np.eye(D), (rp.shape[0], D, D)).copy()
            g[:, a, a] = np.cos(rp[:, ind]); g[:, b, b] = g[:, a, a]
            g[:, a, b] = np.sin(rp[:, ind]); g[:, b, a] = -g[:, a, b]
            q = np.matmul(g, q)
        return "matrix", q
    if spec.rotation_mode == "cayley":                                           # (:793-798)
        t = rp[:, 0]
        m = 1.0 / (1.0 + t ** 2)
        q = np.zeros((rp.shape[0], 2, 2))
        q[:, 0, 0] = q[:, 1, 1] = (1.0 - t ** 2) * m
        q[:, 0, 1] = -2.0 * t * m
        q[:, 1, 0] = 2.0 * t * m
        return "matrix", q
    nt = D * (D - 1) // 2                                                        # triangular_combination (:715-729, 945-951)
    left, mid, right = rp[:, :nt], rp[:, nt:nt + D - 1], rp[:, nt + D - 1:2 * nt + D - 1]
    diag = np.concatenate([mid, -mid.sum(axis=1, keepdims=True)], axis=1)
    return "triangular", (_unit_lower(D, left), diag, np.transpose(_unit_lower(D, right), (0, 2, 1)))


def rotate(rot, x, inverse):
    """x <- R x (sampling direction, :942-987) or R^-1 x (log-prob direction, :1004-1049)."""
    if rot is None:
        return x
    kind, q = rot
    if kind == "matrix":
        return matvec(q, x, transpose=inverse)
    lower, diag, upper = q
    if not inverse:
        return matvec(lower, matvec(upper, x) * np.exp(diag))
    y = matvec(np.linalg.inv(lower), x) / np.exp(diag)
    return matvec(np.linalg.inv(upper), y)


def unpack(spec, params):
    """_obtain_usable_flow_params (gaussianization_flow.py:699-909) + offset split (euclidean_base.py:42-45)."""
    D, K = spec.D, spec.K
    c = 0
    offset = None
    if spec.model_offset:
        offset = params[:, :D]
        c = D
    Q = None
    if spec.n_rot:
        Q = rotation(spec, params[:, c:c + spec.n_rot])
        c += spec.n_rot
    if spec.stretch == "classic":
        km = K - (1 if spec.center_mean else 0)
        means = params[:, c:c + km * D].reshape(-1, km, D); c += km * D
        logw = params[:, c:c + K * D].reshape(-1, K, D); c += K * D
        if spec.fit_normalization:
            lognorm = params[:, c:c + K * D].reshape(-1, K, D); c += K * D
        else:
            lognorm = np.zeros_like(logw)
        log_skew, signs = None, None
        if spec.add_skewness:                                                    # (:352-368, 832-834, 856-858)
            log_skew = bounded_log_fn(params[:, c:c + K * D].reshape(-1, K, D), 0.1, 9.0, center=True); c += K * D
            signs = np.ones(K)
            signs[K // 2:] = -1.0
        logw = _width_regulator(spec, logw)
        if spec.fit_normalization and spec.regulate_normalization:
            lognorm = bounded_log_fn(lognorm, spec.norm_min, spec.norm_max, center=False)
        if spec.center_mean:                                                     # (:846-852)
            w = np.exp(lognorm)
            last = -(means * w[:, :-1, :]).sum(axis=1, keepdims=True) / w[:, -1:, :]
            means = np.concatenate([means, last], axis=1)
        return offset, Q, (means, logw, lognorm, log_skew, signs)
    lw = params[:, c:c + D * K].reshape(-1, D, K); c += D * K
    lh = params[:, c:c + D * K].reshape(-1, D, K); c += D * K
    ld = params[:, c:c + D * (K + 1)].reshape(-1, D, K + 1); c += D * (K + 1)
    box = params[:, c:c + 4 * D].reshape(-1, D, 4)
    left = box[:, :, 0:1]
    right = left + np.exp(box[:, :, 1:2]) + 0.5
    bottom = box[:, :, 2:3]
    top = bottom + np.exp(box[:, :, 3:4]) + 0.5
    return offset, Q, (lw, lh, ld, left, right, bottom, top)


def log_one_plus_exp_x_to_a_minus_1(x, a):
    """extra_functions.log_one_plus_exp_x_to_a_minus_1 (extra_functions.py:14-61): log(((1+e^x)^a - 1) / (1+e^x)^a), branch for branch."""
    x, a = np.broadcast_arrays(x, a)
    sp = a * softplus(x)
    small = x <= -20
    res = np.where(small, np.log(a) + x, 0.0)
    large = sp > 20
    res = np.where(~small & large, sp, res)
    tiny = sp < 1e-8
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        res = np.where(~small & tiny, np.log(sp), res)
        res = np.where(~small & ~large & ~tiny, np.log(np.exp(np.minimum(sp, 50.0)) - 1.0), res)
    return res - sp


def log_quantities(x, means, logw, lognorm, log_skew=None, signs=None, want_pdf=True):
    """logistic_kernel_log_pdf_quantities (gaussianization_flow.py:389-454)."""
    u = (x[:, None, :] - means) / np.exp(logw)
    ln_pi = lognorm - logsumexp(lognorm, axis=1, keepdims=True)
    if log_skew is not None:
        a = np.exp(log_skew)
        sg = signs[None, :, None]
        pos = np.broadcast_to(sg > 0, u.shape)
        log_pdf = None
        if want_pdf:
            log_pdf = logsumexp(-sg * u - logw + log_skew - (a + 1.0) * softplus(-sg * u) + ln_pi, axis=1)
        log_cdfs = np.where(pos, -a * softplus(-u), log_one_plus_exp_x_to_a_minus_1(u, a))
        log_sfs = np.where(pos, log_one_plus_exp_x_to_a_minus_1(-u, a), -a * softplus(u))
        return logsumexp(log_cdfs + ln_pi, axis=1), logsumexp(log_sfs + ln_pi, axis=1), log_pdf
    sp = softplus(-u)
    log_pdf = None
    if want_pdf:
        log_pdf = logsumexp(-u - logw - 2.0 * sp + ln_pi, axis=1)
    log_sf = logsumexp(-u - sp + ln_pi, axis=1)
    log_cdf = logsumexp(-sp + ln_pi, axis=1)
    return log_cdf, log_sf, log_pdf


def _pade_value(log_cdf, log_sf):
    c = 2.0 / (np.pi * PADE_A)
    ln_fac = log_cdf + log_sf + np.log(4.0)
    comb = c + ln_fac / 2.0
    pos = 2.0 * (np.sqrt(comb ** 2 - ln_fac / PADE_A) - comb)
    pos = np.where(pos <= 0, 0.0, pos)
    return np.sqrt(pos)


def _pade_log_deriv_core(log_cdf, log_sf):
    c = 2.0 / (np.pi * PADE_A)
    ln_fac = log_cdf + log_sf + np.log(4.0)
    F = ln_fac / 2.0 + c
    F2 = np.sqrt(F ** 2 - ln_fac / PADE_A)
    with np.errstate(divide="ignore", invalid="ignore"):
        log_num = np.log(-(F - 1.0 / PADE_A - F2))
        log_den = 0.5 * np.log(8.0) + 0.5 * np.log(F2 - F) + np.log(F2)
    return log_num - log_den


def inv_cdf_value(spec, log_cdf, log_sf):
    """sigmoid_inv_error_pass_given_cdf_sf (gaussianization_flow.py:480-560)."""
    t = spec.inverse_function_type
    if t == "isigmoid":
        return -log_sf + log_cdf
    cdf = np.exp(log_cdf)
    if "partly" in t:
        good = (cdf > PADE_BOUND) & (cdf < 1.0 - PADE_BOUND)
        central = np.sqrt(2.0) * erfinv(2.0 * np.where(good, cdf, 0.5) - 1.0)
        if t == "inormal_partly_crude":
            with np.errstate(invalid="ignore"):
                tail = np.sqrt(-2.0 * (log_sf + log_cdf)) - 0.4717
        else:
            tail = _pade_value(log_cdf, log_sf)
        right = cdf >= 1.0 - PADE_BOUND
        left = cdf <= PADE_BOUND
        return np.where(right, central + tail, np.where(left, central - tail, central))
    tail = _pade_value(log_cdf, log_sf)
    return np.where(cdf <= 0.5, -tail, tail)


def inv_cdf_log_deriv(spec, log_cdf, log_sf, log_pdf):
    """sigmoid_inv_error_pass_log_derivative_given_cdf_sf (gaussianization_flow.py:568-671)."""
    t = spec.inverse_function_type
    if t == "isigmoid":
        return np.logaddexp(-log_sf, -log_cdf) + log_pdf
    cdf = np.exp(log_cdf)
    with np.errstate(divide="ignore", invalid="ignore"):
        sign_term = np.log(np.where(cdf <= 0.5, 1.0 - 2.0 * cdf, -1.0 + 2.0 * cdf))
    if "partly" in t:
        good = (cdf > PADE_BOUND) & (cdf < 1.0 - PADE_BOUND)
        central = np.log(np.sqrt(2 * np.pi)) + erfinv(2.0 * np.where(good, cdf, 0.5) - 1.0) ** 2 + log_pdf
        if t == "inormal_partly_crude":
            ln_fac = log_cdf + log_sf
            with np.errstate(divide="ignore", invalid="ignore"):
                tf = -0.5 * np.log(-2.0 * ln_fac) - log_sf - log_cdf
        else:
            tf = _pade_log_deriv_core(log_cdf, log_sf) - log_sf - log_cdf + sign_term
            tf = np.where((cdf > 0.49999) & (cdf < 0.50001), np.log(2.506628), tf)
        tail = (cdf >= 1.0 - PADE_BOUND) | (cdf <= PADE_BOUND)
        return np.where(tail, tf + log_pdf, central)
    # inormal_full_pade (:638-671)
    full = (cdf < 0.49999) | (cdf > 0.50001)
    val = _pade_log_deriv_core(log_cdf, log_sf) - log_cdf - log_sf + log_pdf + sign_term
    return np.where(full, val, np.log(2.506628) + log_pdf)


def _value(spec, x, mix):
    lc, ls, _ = log_quantities(x, *mix, want_pdf=False)
    return inv_cdf_value(spec, lc, ls)


def _value_and_logderiv(spec, x, mix):
    lc, ls, lp = log_quantities(x, *mix)
    return inv_cdf_value(spec, lc, ls), inv_cdf_log_deriv(spec, lc, ls, lp)


def inverse(spec, x, log_det, params):
    """log-prob direction: euclidean_base.inv_flow_mapping (euclidean_base.py:34-51) + gf_block._inv_flow_mapping
    (gaussianization_flow.py:995-1114)."""
    offset, Q, rest = unpack(spec, params)
    if offset is not None:
        x = x - offset
    x = rotate(Q, x, True)
    if spec.stretch == "classic":
        y, logd = _value_and_logderiv(spec, x, rest)
        return y, log_det + logd.sum(axis=-1), []
    lw, lh, ld, left, right, bottom, top = rest
    y, lad, bins = splines.rqs_linext(x[:, :, None], lw, lh, ld, False, left, right, bottom, top)
    return y[:, :, 0], log_det + lad[:, :, 0].sum(axis=-1), [bins]


def _sel(a, mask):
    if a is None or a.ndim < 3:          # absent skewness / the per-component signs
        return a
    return a[mask] if a.shape[0] > 1 else a


def bisection_newton(spec, z, mix, lower=-1e5, upper=1e5, n_bisect=25, n_newton=20, tol=1e-14):
    """inverse_bisection_n_newton_joint_func_and_grad (bisection_n_newton.py:11-135).
    Returns (x, n_newton_used, n_nonconverged)."""
    hi = np.full_like(z, upper)
    lo = np.full_like(z, lower)
    mid = None
    for _ in range(n_bisect):
        mid = (hi + lo) / 2.0
        f = _value(spec, mid, mix)
        right = (f < z).astype(z.dtype)
        leftp = 1.0 - right
        ok = (np.abs(f - z) <= 1e-6 * np.abs(z)).astype(z.dtype)
        lo = (1.0 - ok) * (right * mid + leftp * lo) + ok * mid
        hi = (1.0 - ok) * (right * hi + leftp * mid) + ok * mid
    prev = mid.copy()
    active = np.ones(z.shape[0], dtype=bool)
    f_eval = np.zeros_like(z)
    it = 0
    for it in range(1, n_newton + 1):
        sub = tuple(_sel(m, active) for m in mix)
        val, logd = _value_and_logderiv(spec, prev[active], sub)
        f_eval = val - z[active]
        upd = f_eval / np.exp(logd)
        new = prev[active] - upd
        new = np.where(np.isfinite(new), new, prev[active])
        prev[active] = new
        still = np.abs(upd).sum(axis=1) >= tol
        idx = np.nonzero(active)[0]
        active[idx] = still
        if not active.any():
            break
    nonconv = int((np.abs(f_eval) > 1e-7).sum())
    return prev, it, nonconv


def forward(spec, z, log_det, params):
    """sampling direction: gf_block._flow_mapping (gaussianization_flow.py:911-989) + euclidean_base.flow_mapping (:53-76)."""
    offset, Q, rest = unpack(spec, params)
    bins = []
    if spec.stretch == "classic":
        x, _, _ = bisection_newton(spec, z, rest)
        _, logd = _value_and_logderiv(spec, x, rest)
        log_det = log_det - logd.sum(axis=-1)
    else:
        lw, lh, ld, left, right, bottom, top = rest
        y, lad, b = splines.rqs_linext(z[:, :, None], lw, lh, ld, True, left, right, bottom, top)
        x = y[:, :, 0]
        log_det = log_det + lad[:, :, 0].sum(axis=-1)
        bins = [b]
    x = rotate(Q, x, False)
    if offset is not None:
        x = x + offset
    return x, log_det, bins
