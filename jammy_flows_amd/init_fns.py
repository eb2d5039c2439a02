"""Data-driven initialisation of the Euclidean blocks: pdf.init_params(data=...).

Host-side counterpart of jammy_flows/extra_functions.py:101-409 (find_init_pars_of_chained_blocks and its loss functions): the layers of an
e-block are walked from the data side towards the base; every layer's parameters are chosen so that it removes what is left of the data's
structure (mean -> offsets, covariance -> 't', principal axes -> the Householder rotation of the first 'g', marginals -> percentile-placed
mixture components), and the data are then pushed through exactly that layer before the next one is looked at.

What runs where: the per-layer *fits* are a few-parameter scipy.optimize problems on (D x D) matrices and percentiles of the data -- host
work in the reference and here; the *data pass* through each initialised 'g' layer (the mixture CDF + inverse-CDF stage over all rows,
gaussianization_flow.py:474-478 called from extra_functions.py:379) runs in the HIP kernel of the layer (jf_gf_chain_inv), like every other
evaluation of the flow -- there is no eager restatement of it in this package.
"""
import numpy
import scipy.linalg
import torch
from scipy.optimize import minimize

from . import _hip
from .layers.euclidean import gaussianization_flow, multivariate_normal


def _householder_matrix(vs):
    """Q = H_0 H_1 ... with H_i = I - 2 v v^T / |v|^2 (gaussianization_flow.py:457-471); vs (n_iter, D) numpy."""
    D = vs.shape[1]
    q = numpy.eye(D)
    for v in vs:
        v = v / numpy.sqrt((v * v).sum())
        q = q @ (numpy.eye(D) - 2.0 * numpy.outer(v, v))
    return q


def _rotation_loss(target_matrix, n_iter):
    """how well a Householder product maps the diagonal unit vector onto where the target matrix maps it (extra_functions.py:101-122)"""
    D = target_matrix.shape[0]
    test_vec = numpy.ones(D) / numpy.sqrt(float(D))
    v2 = target_matrix @ test_vec

    def loss(a):
        return -((_householder_matrix(numpy.reshape(a, (n_iter, D))) @ test_vec) * v2).sum()
    return loss


def _mvn_lower_triangular(D, params, cov_type):
    """L of the 't' layer at its DEFAULT width options, as the reference's fit uses it (extra_functions.py:132-136: a fresh mvn_block(dim,
    cov_type)): log-diagonal = smooth-saturation regulator between 0.01 and 100 (multivariate_normal.py make_log_positive), strictly-lower
    entries in the sub-diagonal order of matrix_fns.py:33-49."""
    def log_width(x):
        ln_max, ln_min = numpy.log(100.0), numpy.log(0.01)
        return numpy.logaddexp(ln_max - numpy.logaddexp(0.0, -x + ln_max), ln_min)

    if cov_type == "diagonal_symmetric":
        return numpy.eye(D) * numpy.exp(log_width(params[0]))
    L = numpy.diag(numpy.exp(log_width(params[:D])))
    if cov_type == "full":
        low = params[D:]
        c = 0
        for ind in range(D - 1):
            off = D - 1 - ind
            for j in range(ind + 1):
                L[j + off, j] = low[c + j]
            c += ind + 1
    return L


def _mvn_loss(target_matrix, cov_type):
    """reverse KL between N(0, target) and N(0, L L^T) (extra_functions.py:124-156)"""
    D = target_matrix.shape[0]
    inverse_target = scipy.linalg.pinv(target_matrix)
    logdet_target = numpy.linalg.slogdet(target_matrix)[1]

    def loss(a):
        L = _mvn_lower_triangular(D, a, cov_type)
        predicted = L @ L.T
        return 0.5 * (numpy.trace(inverse_target @ predicted) - numpy.linalg.slogdet(predicted)[1] + logdet_target - D)
    return loss


def _mvn_whitening_matrix(D, params, cov_type):
    """the matrix that removes the fitted second moments from the data (extra_functions.py:158-177)"""
    L = _mvn_lower_triangular(D, params, cov_type)
    _, sigma, r = scipy.linalg.svd(scipy.linalg.pinv(L @ L.T))
    return numpy.sqrt(sigma) * r


def _gf_data_pass(layer, cur_data, percentiles, log_bw):
    """cur_data -> CDF_normal^-1(CDF_mixture(cur_data)) for the freshly initialised components (extra_functions.py:379): components at the
    percentiles, widths exp(log_bw) taken as they are (NOT passed through the layer's width regulator -- the reference calls
    sigmoid_inv_error_pass_w_params with the raw values), equal weights, the layer's inverse-CDF type and (if any) unit-exponent skewness.
    One launch of the layer's own kernel with a descriptor that encodes exactly that."""
    D, K = layer.dimension, layer.num_kde
    s = _hip.jf_gf_layer()
    s.num_kde = K
    s.hh_iter = 0
    s.model_offset = 0
    s.fit_normalization = 0
    s.regulate_normalization = 0
    s.inverse_function_type = _hip.GF_INV_TYPES[layer.inverse_function_type]
    s.width_mode = _hip.GF_WIDTH_EXP           # width = exp(x) + width_min with a vanishing width_min: the raw log-width is the log-width
    s.clamp_widths = 0
    s.nonlinear_stretch_type = _hip.GF_STRETCH_CLASSIC
    s.rotation_mode = 0
    s.center_mean = 0
    s.add_skewness = 1 if layer.add_skewness else 0
    s.width_min, s.width_max, s.norm_min, s.norm_max = 1e-300, -1.0, 1.0, 10.0
    parts = [percentiles.reshape(-1), log_bw.reshape(-1)]
    if layer.add_skewness:
        # Reference quirk, reproduced: the pass hands exponent_regulator(0).exp() = 1.0 to a function that takes LOG-exponents
        # (extra_functions.py:372 -> gaussianization_flow.py:389-401), so its components are skewed with exponent e, not 1.  The raw value whose
        # regulated log-exponent is 1:  LSE(ln 9 - softplus(ln 9 - r), ln 0.1) = 1  <=>  r = -ln((9 / (e - 0.1) - 1) / 9)
        raw = -numpy.log((9.0 / (numpy.e - 0.1) - 1.0) / 9.0)
        parts.append(torch.full_like(log_bw, raw).reshape(-1))
    row = torch.cat(parts).reshape(1, -1).to(cur_data)
    y, _ = _hip.gf_chain("inv", cur_data, torch.zeros(cur_data.shape[0], dtype=cur_data.dtype, device=cur_data.device), row,
                         _hip.gf_layer_array([s]), 1, D)
    return y


def find_init_pars_of_chained_blocks(layer_list, data, mvn_min_max_sv_ratio=1e-4):
    """initial parameter vector of one e-block (layer order 0..n-1), from data (B, D) or, without data, the layers' default inits
    (extra_functions.py:179-409).  `data` is moved to the GPU in float64 for the passes through the initialised layers."""
    if data is None:
        return torch.cat([l.get_desired_init_parameters() for l in layer_list]) if len(layer_list) else torch.zeros(0)
    dev = data.device if data.is_cuda else torch.device("cuda")
    out_dtype = data.dtype
    cur = data.detach().to(device=dev, dtype=torch.float64).contiguous()
    B, D = cur.shape
    per_layer = []
    with torch.no_grad():
        for layer_ind, layer in enumerate(layer_list[::-1]):
            plist = []
            if layer.model_offset:
                means = cur.mean(dim=0, keepdim=True)
                plist.append(means.squeeze(0).cpu())
                cur = cur - means
            if type(layer) is multivariate_normal.mvn_block:
                if layer.cov_type == "identity":
                    if plist:
                        per_layer.append(torch.cat(plist))
                    continue
                data_matrix = (cur.T @ cur / float(B)).cpu().numpy()
                l, sigma, r = scipy.linalg.svd(data_matrix)                # lift tiny singular values: keeps the fit well conditioned
                sigma = numpy.where(sigma < mvn_min_max_sv_ratio * sigma.max(), mvn_min_max_sv_ratio * sigma.max(), sigma)
                n_mat = layer.total_param_num - (D if layer.model_offset else 0)
                res = minimize(_mvn_loss((l * sigma) @ r, layer.cov_type), numpy.random.normal(size=n_mat))
                plist.append(torch.from_numpy(res["x"]))
                w = torch.from_numpy(_mvn_whitening_matrix(D, res["x"], layer.cov_type)).to(cur)
                cur = cur @ w.T
            elif type(layer) is gaussianization_flow.gf_block:
                if layer.rotation_mode == "householder":
                    if layer.use_householder:
                        n_iter = layer.householder_iter
                        if D < 30 and layer_ind == 0:
                            # PCA on the layer next to the data: fit the reflections to the right singular vectors of X^T X
                            _, _, r = scipy.linalg.svd((cur.T @ cur).cpu().numpy())
                            res = minimize(_rotation_loss(r, n_iter), numpy.random.normal(size=D * n_iter))
                            vs = res["x"]
                        else:
                            vs = torch.randn(D * n_iter).double().numpy()
                        plist.append(torch.from_numpy(vs))
                        q = torch.from_numpy(_householder_matrix(vs.reshape(n_iter, D))).to(cur)
                        cur = cur @ q                                      # rows: (Q^T x)^T = x^T Q
                else:                                                      # identity rotations for the other parametrisations (:319-345)
                    plist.append(torch.zeros(layer.num_triangle_params + layer.num_angle_pars + layer.num_cayley_pars))
                K = layer.num_kde
                assert K < 100
                if layer.nonlinear_stretch_type != "classic":
                    raise Exception("Data initilaization only implemented (and probably only makes sense) for classic Gaussianization Flow structure")
                pct = torch.from_numpy(numpy.percentile(cur.cpu().numpy(), numpy.linspace(0, 100, K), axis=0)).to(cur)    # (K, D)
                plist.append((pct if not layer.center_mean else pct[:-1]).reshape(-1).cpu())
                log_bw = torch.log((pct[1:] - pct[:-1]).min(dim=0, keepdim=True)[0] * 1.5) * torch.ones_like(pct)         # (K, D)
                plist.append(log_bw.reshape(-1).cpu())
                if layer.fit_normalization:
                    plist.append(torch.ones(K * D, dtype=torch.float64))
                if layer.add_skewness:
                    plist.append(torch.zeros(K * D, dtype=torch.float64))
                cur = _gf_data_pass(layer, cur, pct, log_bw)
            else:
                plist.append(layer._get_desired_init_parameters().double())
            per_layer.append(torch.cat([p.double() for p in plist]) if plist else torch.zeros(0, dtype=torch.float64))
    params = torch.cat(per_layer[::-1]).to(out_dtype) if per_layer else torch.zeros(0, dtype=out_dtype)
    expected = sum(l.total_param_num for l in layer_list)
    assert len(params) == expected, "Total number of defined params (%d) does not match expected params based on layer definitions (%d)" % (
        len(params), expected)
    return params
