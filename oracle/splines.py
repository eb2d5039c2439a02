"""Rational-quadratic spline family of the reference (layers/spline_fns.py) restated in numpy.

Every function returns (outputs, logabsdet, bin_idx) -- bin_idx (int64) is what the bit-exact parity tests pin.
Oracle = test infrastructure only.
"""
import numpy as np

from .special import softplus

TWO_PI = 2.0 * np.pi


def _softmax(a):
    a = a - np.max(a, axis=-1, keepdims=True)
    e = np.exp(a)
    return e / np.sum(e, axis=-1, keepdims=True)


def _sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def searchsorted(knots, inputs, eps=1e-6):
    """spline_fns.py:13-19: last knot bumped by eps, bin = #(x >= knot) - 1  (SURVEY D7)."""
    k = knots.copy()
    k[..., -1] += eps
    return (inputs >= k).sum(axis=-1, keepdims=True).astype(np.int64) - 1


def _cum_knots(unnormalized, lo, hi, rel_min, pin_ends):
    """softmax -> min-size mix -> cumsum -> affine to [lo, hi]  (spline_fns.py:88-98 / 220-235)."""
    nb = unnormalized.shape[-1]
    frac = rel_min + (1.0 - rel_min * nb) * _softmax(unnormalized)
    cum = np.cumsum(frac, axis=-1)
    cum = np.concatenate([np.zeros_like(cum[..., :1]), cum], axis=-1)
    cum = (hi - lo) * cum + lo
    if pin_ends:
        cum[..., 0] = lo
        cum[..., -1] = hi
    return cum, cum[..., 1:] - cum[..., :-1]


def _restrict(unw, unh, ratio):
    """restrict_max_min_width_height_ratio (spline_fns.py:80-85)."""
    if ratio > 0.0:
        nb = unw.shape[-1]
        ln_max = (np.log(ratio) - np.log(nb - 1)) / 2.0
        assert ln_max > 0
        unw = 2.0 * _sigmoid(unw) * ln_max - ln_max
        unh = 2.0 * _sigmoid(unh) * ln_max - ln_max
    return unw, unh


def _gather(arr, idx):
    arr = np.broadcast_to(arr, idx.shape[:-1] + arr.shape[-1:])
    return np.take_along_axis(arr, idx, axis=-1)


def _rq_core(inputs, cumw, w, cumh, h, d, bin_idx, inverse):
    """closed-form forward / quadratic-root inverse of one RQ bin (spline_fns.py:127-186)."""
    in_cumw = _gather(cumw, bin_idx)
    in_w = _gather(w, bin_idx)
    in_cumh = _gather(cumh, bin_idx)
    in_delta = _gather(h / w, bin_idx)
    in_d = _gather(d, bin_idx)
    in_d1 = _gather(d[..., 1:], bin_idx)
    in_h = _gather(h, bin_idx)
    if inverse:
        dy = inputs - in_cumh
        s = in_d + in_d1 - 2.0 * in_delta
        a = dy * s + in_h * (in_delta - in_d)
        b = in_h * in_d - dy * s
        c = -in_delta * dy
        disc = b * b - 4.0 * a * c
        root = (2.0 * c) / (-b - np.sqrt(disc))
        out = root * in_w + in_cumw
        t1mt = root * (1.0 - root)
        den = in_delta + s * t1mt
        num = in_delta ** 2 * (in_d1 * root ** 2 + 2.0 * in_delta * t1mt + in_d * (1.0 - root) ** 2)
        lad = np.log(num) - 2.0 * np.log(den)
        return out, -lad, disc
    theta = (inputs - in_cumw) / in_w
    t1mt = theta * (1.0 - theta)
    s = in_d + in_d1 - 2.0 * in_delta
    numer = in_h * (in_delta * theta ** 2 + in_d * t1mt)
    den = in_delta + s * t1mt
    out = in_cumh + numer / den
    num = in_delta ** 2 * (in_d1 * theta ** 2 + 2.0 * in_delta * t1mt + in_d * (1.0 - theta) ** 2)
    lad = np.log(num) - 2.0 * np.log(den)
    return out, lad, None


def rqs_plain(inputs, unw, unh, und, inverse, left, right, bottom, top,
              min_w=1e-3, min_h=1e-3, min_d=1e-3, ratio=-1.0):
    """rational_quadratic_spline (spline_fns.py:45-186).  inputs (B,1); params (PB,nb)/(PB,nb+1)."""
    if inputs.min() < left or inputs.max() > right:
        raise ValueError("outside boundaries in rational-spline flow")
    unw, unh = _restrict(unw, unh, ratio)
    cumw, w = _cum_knots(unw, left, right, min_w, True)
    cumh, h = _cum_knots(unh, bottom, top, min_h, True)
    d = min_d + softplus(und)
    bin_idx = searchsorted(cumh if inverse else cumw, inputs)
    out, lad, disc = _rq_core(inputs, cumw, w, cumh, h, d, bin_idx, inverse)
    if inverse:
        assert (disc >= 0).all()
    return out, lad, bin_idx


def rqs_linext(inputs, unw, unh, und, inverse, left, right, bottom, top,
               min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """rational_quadratic_spline_with_linear_extension (spline_fns.py:188-358).
    inputs (B,D,1); params (PB,D,nb); box (PB,D,1).  eps=0 search + index clamp + linear tails."""
    cumw, w = _cum_knots(unw, left, right, min_w, False)
    cumh, h = _cum_knots(unh, bottom, top, min_h, False)
    d = min_d + softplus(und)
    raw_idx = searchsorted(cumh if inverse else cumw, inputs, eps=0.0)
    nb = h.shape[-1]
    bin_idx = np.clip(raw_idx, 0, nb - 1)          # spline_fns.py:257-258 ("keep bin idx sane")
    with np.errstate(invalid="ignore", divide="ignore"):
        out, lad, _ = _rq_core(inputs, cumw, w, cumh, h, d, bin_idx, inverse)
    d0, dl = d[..., 0:1], d[..., -1:]
    if inverse:
        lo_off = cumw[..., 0:1] - cumh[..., 0:1] / d0
        out = np.where(inputs <= bottom, inputs / d0 + lo_off, out)
        hi_off = cumw[..., -1:] - cumh[..., -1:] / dl
        out = np.where(inputs >= top, inputs / dl + hi_off, out)
        lad = np.where(inputs <= bottom, -np.log(d0), lad)
        lad = np.where(inputs >= top, -np.log(dl), lad)
    else:
        lo_off = cumh[..., 0:1] - cumw[..., 0:1] * d0
        out = np.where(inputs <= left, inputs * d0 + lo_off, out)
        hi_off = cumh[..., -1:] - cumw[..., -1:] * dl
        out = np.where(inputs >= right, inputs * dl + hi_off, out)
        lad = np.where(inputs <= left, np.log(d0), lad)
        lad = np.where(inputs >= right, np.log(dl), lad)
    return out, lad, raw_idx                        # raw (-1 .. nb) count, as spline_fns.searchsorted returns it


def rqs_smooth(inputs, unw, unh, un_bd, inverse, left, right, bottom, top,
               min_w=1e-3, min_h=1e-3, min_d=1e-3, ratio=-1.0, solution_index=0):
    """rational_quadratic_spline_smooth (spline_fns.py:361-558): interior derivatives in closed form (2 or 3 bins)."""
    if inputs.min() < left or inputs.max() > right:
        raise ValueError("outside boundaries in rational-spline flow")
    assert un_bd.shape[-1] == 2
    unw, unh = _restrict(unw, unh, ratio)
    cumw, w = _cum_knots(unw, left, right, min_w, True)
    cumh, h = _cum_knots(unh, bottom, top, min_h, True)
    bd = min_d + softplus(un_bd)
    nb = w.shape[-1]
    if nb == 2:
        hs = h[..., :-1] + h[..., 1:]
        lo_p = h[..., :-1] / hs
        hi_p = h[..., 1:] / hs
        neg_p_half = 0.5 * (lo_p * ((h[..., 1:] / w[..., 1:]) - bd[..., 1:]) + hi_p * ((h[..., :-1] / w[..., :-1]) - bd[..., :-1]))
        q = -(h[..., :-1] * h[..., 1:]) * (lo_p * (1.0 / w[..., :-1] ** 2) + hi_p * (1.0 / w[..., 1:] ** 2))
        sq = np.sqrt(neg_p_half ** 2 - q)
        res = neg_p_half + sq if solution_index == 0 else neg_p_half - sq
        d = np.concatenate([bd[..., :1], res, bd[..., 1:]], axis=-1)
    elif nb == 3:
        w1, w2, h1, h2 = w[..., 0:1], w[..., 1:2], h[..., 0:1], h[..., 1:2]
        cden = w1 * w2 * (2 * h1 + h2)
        p = h2 * (bd[..., :1] * w1 * w2 - h1 * (w1 + w2)) / cden
        q = -h1 * h2 * (h1 * w2 ** 2 + h2 * w1 ** 2) / (cden * w1 * w2)
        neg_p_half = -p / 2.0
        res = neg_p_half + np.sqrt(neg_p_half ** 2 - q)
        d = np.concatenate([bd[..., :1], res, res, bd[..., 1:]], axis=-1)
    elif nb == 1:
        d = bd
    else:
        raise NotImplementedError
    bin_idx = searchsorted(cumh if inverse else cumw, inputs)
    out, lad, disc = _rq_core(inputs, cumw, w, cumh, h, d, bin_idx, inverse)
    if inverse:
        assert (disc >= 0).all()
    return out, lad, bin_idx


def rqs_circular(inputs, unw, unh, inverse, min_w=1e-3, min_h=1e-3, ratio=-1.0, shift_to_middle=True):
    """rational_quadratic_spline_smooth_circular (spline_fns.py:561-760): periodic 2-bin spline on [0, 2pi]."""
    left, right, bottom, top = 0.0, TWO_PI, 0.0, TWO_PI
    if inputs.min() < left or inputs.max() > right:
        raise ValueError("outside boundaries in rational-spline flow")
    unw, unh = _restrict(unw, unh, ratio)
    cumw, w = _cum_knots(unw, left, right, min_w, True)
    cumh, h = _cum_knots(unh, bottom, top, min_h, True)
    assert w.shape[-1] == 2
    w1, w2, h1, h2 = w[..., :1], w[..., 1:], h[..., :1], h[..., 1:]
    hp, wp = h1 * h2, w1 * w2
    sq = np.sqrt(hp * (8 * ((h2 * w1) ** 2 + (h1 * w2) ** 2) + (9 * (w1 + w2) ** 2 - 16 * wp) * hp))
    res = (hp * (w1 + w2) + sq) / (4 * (h1 + h2) * wp)
    d = np.concatenate([res, res, res], axis=-1)
    corr = 0.0
    if shift_to_middle:
        w1mx = -np.pi + w1 / 2.0
        w1mx_p_w2 = w1mx + w2
        nom = h2 * w1mx * (w1mx * h1 - res * w1 * w1mx_p_w2)
        den = h1 * w2 ** 2 + 2 * (h1 - res * w1) * w1mx * w1mx_p_w2
        corr = TWO_PI - (h1 + nom / den)
    used = inputs
    if shift_to_middle:
        if inverse:
            used = inputs - corr
        else:
            used = inputs - (np.pi - w[..., 0:1] / 2.0)
        used = np.where(used < 0.0, used + TWO_PI, used)
    bin_idx = searchsorted(cumh if inverse else cumw, used)
    out, lad, disc = _rq_core(used, cumw, w, cumh, h, d, bin_idx, inverse)
    if inverse:
        assert (disc >= 0).all()
    if shift_to_middle:
        out = out + ((np.pi - w[..., 0:1] / 2.0) if inverse else corr)
        out = np.where(out > TWO_PI, out - TWO_PI, out)
        out = np.where(inputs == 0.0, 0.0, out)
        out = np.where(inputs == TWO_PI, TWO_PI, out)
    return out, lad, bin_idx
