"""'t' affine flow / multivariate normal, restated in numpy.  Oracle = test infrastructure only.

  mvn_block._inv_flow_mapping / _flow_mapping        jammy_flows/layers/euclidean/multivariate_normal.py:226-263
  make_log_positive (width regulators)               :113-157 (same functions as the 'g' layer's, gaussianization_flow.py:269-317)
  lower-triangular matrix from the parameter row     jammy_flows/layers/matrix_fns.py:4-52  (sub-diagonals from the bottom-left corner)
  explicit inverse (the reference multiplies by it)  matrix_fns.py:54-145  -- here numpy.linalg.inv of the same matrix
  offset                                             jammy_flows/layers/euclidean/euclidean_base.py:34-76
"""
import numpy as np

from .gf import _width_regulator


class TSpec:
    def __init__(self, dimension, opts, model_offset):
        self.D = dimension
        self.cov_type = opts["cov_type"]
        self.model_offset = model_offset
        self.softplus_for_width = opts["softplus_for_width"]
        self.width_smooth_saturation = opts["width_smooth_saturation"]
        self.clamp_widths = opts["clamp_widths"]
        self.width_min = opts["lower_bound_for_widths"]
        self.width_max = opts["upper_bound_for_widths"] if opts["upper_bound_for_widths"] > 0 else None
        D = dimension
        own = {"identity": 0, "diagonal_symmetric": 1, "diagonal": D, "full": D + D * (D - 1) // 2}[self.cov_type]
        self.total_param_num = own + (D if model_offset else 0)

    def row_from_state(self, sd, prefix):
        parts = []
        if self.model_offset:
            parts.append(np.asarray(sd[prefix + "offsets"], dtype=np.float64).reshape(-1))
        if self.cov_type == "diagonal_symmetric":
            parts.append(np.asarray(sd[prefix + "single_diagonal_log"], dtype=np.float64).reshape(-1))
        elif self.cov_type in ("diagonal", "full"):
            parts.append(np.asarray(sd[prefix + "full_diagonal_log"], dtype=np.float64).reshape(-1))
            if self.cov_type == "full":
                parts.append(np.asarray(sd[prefix + "lower_triangular_entries"], dtype=np.float64).reshape(-1))
        return np.concatenate(parts).reshape(1, -1) if parts else np.zeros((1, 0))


def _matrix(spec, own):
    """(B|1, D, D) lower-triangular matrix and (B|1,) log-determinant (matrix_fns.py:4-52)"""
    D = spec.D
    if spec.cov_type == "diagonal_symmetric":
        s = _width_regulator(spec, own[:, :1])
        return np.exp(s)[:, :, None] * np.eye(D)[None], D * s[:, 0]
    s = _width_regulator(spec, own[:, :D])
    L = np.zeros((own.shape[0], D, D))
    L[:, np.arange(D), np.arange(D)] = np.exp(s)
    if spec.cov_type == "full":
        low = own[:, D:]
        cum = np.cumsum(np.arange(D) + 1)
        for ind in range(D - 1):
            k = D - 1 - ind                                   # sub-diagonal i - j = k holds ind + 1 entries
            seg = low[:, (cum[ind - 1] if ind > 0 else 0):cum[ind]]
            for t in range(ind + 1):
                L[:, t + k, t] = seg[:, t]
    return L, s.sum(axis=1)


def inverse(spec, x, log_det, params):
    c = 0
    if spec.model_offset:
        x = x - params[:, :spec.D]
        c = spec.D
    if spec.cov_type == "identity":
        return x, log_det, []
    L, ld = _matrix(spec, params[:, c:])
    z = np.einsum("bij,bj->bi", np.broadcast_to(np.linalg.inv(L), (x.shape[0],) + L.shape[1:]), x)
    return z, log_det - ld, []


def forward(spec, z, log_det, params):
    c = spec.D if spec.model_offset else 0
    if spec.cov_type == "identity":
        x = z
    else:
        L, ld = _matrix(spec, params[:, c:])
        x = np.einsum("bij,bj->bi", np.broadcast_to(L, (z.shape[0],) + L.shape[1:]), z)
        log_det = log_det + ld
    if spec.model_offset:
        x = x + params[:, :spec.D]
    return x, log_det, []
