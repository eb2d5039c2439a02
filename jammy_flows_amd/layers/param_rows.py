"""Helpers shared by the host-side layer classes: cached flat parameter rows and spline option bookkeeping."""
import numpy
import torch

from .. import _hip


class PermanentRowCache:
    """the permanent nn.Parameters of a layer as ONE (1, total_param_num) row in the reference's extra_inputs layout,
    rebuilt only when a parameter changed (version counter / storage pointer)."""

    def __init__(self):
        self._cache = None

    def get(self, tensors, like, expected_len):
        key = (like.dtype, like.device, tuple((t._version, t.data_ptr()) for t in tensors))
        if self._cache is None or self._cache[0] != key:
            with torch.no_grad():
                if len(tensors) == 0:
                    row = torch.zeros((1, 0), dtype=like.dtype, device=like.device)
                else:
                    row = torch.cat([t.detach().reshape(-1).to(device=like.device, dtype=like.dtype) for t in tensors]).reshape(1, -1)
            assert row.shape[1] == expected_len, (row.shape, expected_len)
            self._cache = (key, row)
        return self._cache[1]


def spline_counts(num_basis_functions, fix_first, fix_second, smooth, fix_boundary_derivatives, min_derivative, circular):
    """(n_w, n_h, n_d, fixed-boundary value) of 'r' (rational_quadratic_spline.py:99-170) and 'o' (splines_1d.py:39-101)."""
    nb = num_basis_functions
    n_w = n_h = nb
    if fix_first:
        n_w = n_h = nb - 1
        if fix_second > 0:
            n_w -= 1
    fixed_val = 0.0
    fix_bd = 1 if fix_boundary_derivatives > 0.0 else 0
    if fix_bd:
        fixed_val = float(numpy.log(numpy.exp(fix_boundary_derivatives - min_derivative) - 1.0))
    if not circular:
        if smooth == 1:
            assert nb in (2, 3), "Only support 2/3 basis functions for smooth derivative!"
            sub = {2: (3 if fix_bd else 1), 3: (4 if fix_bd else 2)}[nb]
        else:
            sub = 2 if fix_bd else 0
            if fix_bd:
                assert fix_boundary_derivatives > min_derivative
    else:
        if smooth == 1:
            assert nb == 2, "Only support 2 basis functions for smooth derivative!"
            sub = 3
        elif fix_bd:
            sub = 2
            assert fix_boundary_derivatives > min_derivative, "Fixed boundary derivative should be larger than min derivative!"
        else:
            sub = 1
    n_d = nb + 1 - sub
    if smooth and nb == 3:
        n_w -= 1
        n_h -= 1
    return n_w, n_h, n_d, fix_bd, fixed_val


def spline_struct(layer, ratio=-1.0):
    s = _hip.jf_spline_opts()
    s.num_bins = layer.num_basis_functions
    s.smooth = 1 if layer.smooth_second_derivative else 0
    s.fix_first = 1 if layer.fix_first_width_n_height_to_zero else 0
    s.fix_second = 1 if layer.also_fix_second_width_to_zero else 0
    s.independent = 1 if layer.independent_width_height_parametrization else 0
    s.fix_bd = layer._fix_bd
    s.n_w, s.n_h, s.n_d = layer.num_width_params, layer.num_height_params, layer.num_derivative_params
    s.fix_bd_value = layer._fix_bd_value
    s.min_w, s.min_h, s.min_d = float(layer.min_width), float(layer.min_height), float(layer.min_derivative)
    s.ratio = float(ratio)
    if s.num_bins > _hip.JF_SPLINE_CAP:
        raise NotImplementedError("splines with more than %d bins have no HIP kernel" % _hip.JF_SPLINE_CAP)
    return s
