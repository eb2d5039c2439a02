"""Interval / sphere base-class arithmetic of the reference restated in numpy:
layers/intervals/interval_base.py, layers/spheres/sphere_base.py.  Oracle = test infrastructure only."""
import numpy as np

from .special import LN_2PI, erf, erfinv, householder_matrix

PI = np.pi
TWO_PI = 2.0 * np.pi


# ------------------------------------------------------------------ intervals
def real_line_to_interval(x, log_det, lo, hi):
    """interval_base.py:33-45."""
    w = hi - lo
    res = (0.5 + 0.5 * erf(x / np.sqrt(2.0))) * w + lo
    return res, log_det - (x[:, 0] ** 2) / 2.0 - 0.5 * LN_2PI + np.log(w)


def interval_to_real_line(x, log_det, lo, hi):
    """interval_base.py:47-59."""
    w = hi - lo
    with np.errstate(divide="ignore"):
        res = erfinv(2.0 * ((x - lo) / w) - 1.0) * np.sqrt(2.0)
    return res, log_det - (-(res[:, 0] ** 2) / 2.0 - 0.5 * LN_2PI + np.log(w))


# ------------------------------------------------------------------ spheres: clamps
def safe_angle_within_pi(x, margin=1e-7):
    """sphere_base.py:8-19."""
    return np.where(x > PI - margin, PI - margin, np.where(x < margin, margin, x))


def safe_costheta(x, margin=1e-10):
    """sphere_base.py:21-38 (float64 default margin 1e-10)."""
    return np.where(x > 1.0 - margin, 1.0 - margin, np.where(x < -1.0 + margin, -1.0 + margin, x))


def safe_angle_within_2pi(x, margin=1e-7):
    """spline_fns.py:22-43."""
    return np.where(x > TWO_PI - margin, TWO_PI - margin, np.where(x < margin, margin, x))


# ------------------------------------------------------------------ spheres: embeddings
def spherical_to_eucl(x, log_det, dim):
    """sphere_base.spherical_to_eucl_embedding (sphere_base.py:305-335)."""
    if dim == 1:
        return np.concatenate([np.cos(x), np.sin(x)], axis=1), log_det
    theta = safe_angle_within_pi(x[:, 0:1])
    phi = x[:, 1:2]
    e = np.concatenate([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=1)
    return e, log_det + np.log(np.sin(theta)).sum(axis=-1)


def eucl_to_spherical(x, log_det, dim):
    """sphere_base.eucl_to_spherical_embedding (sphere_base.py:242-303)."""
    if dim == 1:
        r = np.sqrt((x ** 2).sum(axis=1, keepdims=True))
        ang = np.arccos(x[:, 0:1] / r)
        ang = np.where(x[:, 1:2] < 0, TWO_PI - ang, ang)
        return ang, log_det
    theta = np.arccos(x[:, 2:3] / np.sqrt((x ** 2).sum(axis=-1, keepdims=True)))
    theta = safe_angle_within_pi(theta)
    log_det = log_det - np.log(np.sin(theta)).sum(axis=-1)
    arg = x[:, 0:1] / np.sqrt((x[:, :2] ** 2).sum(axis=-1, keepdims=True))
    arg = np.clip(arg, -1.0, 1.0)
    phi = np.arccos(arg)
    phi = np.where(x[:, 1:2] < 0, TWO_PI - phi, phi)
    return np.concatenate([theta, phi], axis=1), log_det


# ------------------------------------------------------------------ spheres: charts to the plane
def sphere_to_plane(x, log_det, dim):
    """sphere_base.sphere_to_plane + inplane_spherical_to_euclidean (sphere_base.py:456-521, 410-430), float64 eps."""
    if dim == 1:
        sign = np.where(x > PI, -1.0, 1.0)
        nx = np.where(sign > 0, x, TWO_PI - x)
        eps = 1e-8
        nx = np.where(nx <= 0.0, eps, nx)
        nx = np.where(nx >= TWO_PI, TWO_PI - eps, nx)
        y = np.sqrt(2.0) * erfinv(1.0 - nx / PI)
        log_det = log_det - np.log(np.sqrt(TWO_PI)) + (y[:, 0] ** 2) / 2.0
        return y * sign, log_det
    st = safe_angle_within_pi(x[:, 0:1])
    c = safe_costheta(np.cos(st), margin=1e-6)
    r = np.sqrt(-np.log((1.0 - c) / 2.0) * 2.0)
    log_det = log_det - np.log(1.0 - c[:, 0]) + np.log(np.sin(st[:, 0]))
    phi = x[:, 1:2]
    return np.concatenate([r * np.cos(phi), r * np.sin(phi)], axis=1), log_det


def plane_to_sphere(x, log_det, dim):
    """sphere_base.plane_to_sphere + inplane_euclidean_to_spherical (sphere_base.py:523-598, 364-408)."""
    radius = np.sqrt((x ** 2).sum(axis=1, keepdims=True))
    if dim == 1:
        keep = (x >= 0) * 1.0
        log_det = log_det + np.log(np.sqrt(TWO_PI)) - (radius[:, 0] ** 2) / 2.0
        a = PI * (1.0 - erf(radius / np.sqrt(2.0)))
        return keep * a + (1.0 - keep) * (TWO_PI - a), log_det
    with np.errstate(invalid="ignore", divide="ignore"):
        arg = np.where(radius == 0, 1.0, x[:, :1] / radius)
    ang = np.arccos(arg)
    ang = np.where(x[:, 1:2] < 0, TWO_PI - ang, ang)
    theta = np.arccos(1.0 - 2.0 * np.exp(-(radius ** 2) / 2.0))
    theta = safe_angle_within_pi(theta)
    log_det = log_det + np.log(1.0 - np.cos(theta[:, 0])) - np.log(np.sin(theta[:, 0]))
    return np.concatenate([theta, ang], axis=1), log_det


# ------------------------------------------------------------------ spheres: rotations
def rotation_matrix(rot_params, dim, n_iter, mode="householder"):
    """sphere_base.compute_rotation_matrix (sphere_base.py:112-216): householder / angles (Givens) / xyz / quaternion."""
    if mode == "householder":
        return householder_matrix(rot_params.reshape(-1, n_iter, dim + 1))
    B, E = rot_params.shape[0], dim + 1
    if mode == "angles":
        R = np.broadcast_to(np.eye(E), (B, E, E)).copy()
        ind = 0
        for a in range(E):
            for b in range(a + 1, E):
                G = np.broadcast_to(np.eye(E), (B, E, E)).copy()
                c, s = np.cos(rot_params[:, ind]), np.sin(rot_params[:, ind])
                G[:, a, a] = c; G[:, b, b] = c; G[:, a, b] = s; G[:, b, a] = -s
                R = G @ R
                ind += 1
        return R
    R = np.zeros((B, 3, 3))
    if mode == "xyz":
        n = rot_params / np.sqrt((rot_params ** 2).sum(axis=1, keepdims=True))
        mx, my, mz = n[:, 0], n[:, 1], n[:, 2]
        d = 1.0 + mz
        R[:, 0, 0] = 1 - mx * mx / d; R[:, 0, 1] = -mx * my / d; R[:, 0, 2] = mx
        R[:, 1, 0] = -mx * my / d; R[:, 1, 1] = 1 - my * my / d; R[:, 1, 2] = my
        R[:, 2, 0] = -mx; R[:, 2, 1] = -my; R[:, 2, 2] = mz
        return R
    assert mode == "quaternion"
    a, i, j, k = (rot_params[:, q] for q in range(4))
    n2 = (rot_params ** 2).sum(axis=1)
    R[:, 0, 0] = 1 - 2 * (j * j + k * k) / n2; R[:, 0, 1] = 2 * (i * j - a * k) / n2; R[:, 0, 2] = 2 * (i * k + j * a) / n2
    R[:, 1, 0] = 2 * (i * j + a * k) / n2; R[:, 1, 1] = 1 - 2 * (i * i + k * k) / n2; R[:, 1, 2] = 2 * (j * k - i * a) / n2
    R[:, 2, 0] = 2 * (i * k - j * a) / n2; R[:, 2, 1] = 2 * (j * k + i * a) / n2; R[:, 2, 2] = 1 - 2 * (i * i + j * j) / n2
    return R
