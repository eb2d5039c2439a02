"""Elementwise helpers with the semantics of the torch ops the reference calls (oracle = test infrastructure)."""
import numpy as np
from scipy import special as _sp

erf = _sp.erf
erfinv = _sp.erfinv

LN_2PI = float(np.log(2.0 * np.pi))


def softplus(x):
    """torch.nn.functional.softplus(beta=1, threshold=20): identity above the threshold."""
    x = np.asarray(x)
    with np.errstate(over="ignore"):
        return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


def logsumexp(a, axis, keepdims=False):
    """torch.logsumexp: max-shifted; an all -inf slice gives -inf."""
    a = np.asarray(a)
    m = np.max(a, axis=axis, keepdims=True)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        s = np.log(np.sum(np.exp(a - m_safe), axis=axis, keepdims=True)) + m_safe
    return s if keepdims else np.squeeze(s, axis=axis)


def logaddexp(a, b):
    return np.logaddexp(a, b)


def bounded_log_fn(x, min_val, max_val, center):
    """generate_log_function_bounded_in_logspace (layers/euclidean/gaussianization_flow.py:23-47):
    soft clamp of a log-quantity between ln(min_val) and ln(max_val)."""
    ln_max = np.log(max_val)
    ln_min = np.log(min_val)
    center_val = ln_max if center else 0.0
    first = ln_max - np.logaddexp(0.0, -x + center_val)
    return np.logaddexp(first, ln_min)


def normal_logpdf_sum(z):
    """torch.distributions.Normal(0,1).log_prob(z).sum(-1)  (main/default.py:1110-1115)."""
    return (-0.5 * z * z - 0.5 * LN_2PI).sum(axis=-1)


def householder_matrix(vs):
    """Q = H_0 H_1 ... with H_i = I - 2 v v^T / |v|^2 (gaussianization_flow.py:457-471, sphere_base.py:222-240).
    vs: (PB, n_iter, dim) -> (PB, dim, dim)."""
    pb, n_iter, dim = vs.shape
    q = np.broadcast_to(np.eye(dim, dtype=vs.dtype), (pb, dim, dim)).copy()
    for i in range(n_iter):
        v = vs[:, i, :]
        v = v / np.sqrt((v * v).sum(axis=1, keepdims=True))
        qi = np.eye(dim, dtype=vs.dtype)[None] - 2.0 * v[:, :, None] * v[:, None, :]
        q = np.matmul(q, qi)
    return q


def matvec(m, x, transpose=False):
    """batched (or broadcast, PB=1) matrix-vector product."""
    if transpose:
        return np.einsum("bji,bj->bi", np.broadcast_to(m, (x.shape[0],) + m.shape[1:]), x)
    return np.einsum("bij,bj->bi", np.broadcast_to(m, (x.shape[0],) + m.shape[1:]), x)
