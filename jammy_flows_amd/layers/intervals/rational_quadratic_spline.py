"""Rational-quadratic spline layer 'r' on an interval -- host side.

Constructor arguments, parameter names / shapes and bookkeeping of
jammy_flows/layers/intervals/rational_quadratic_spline.py (:61-178, 403-450); the arithmetic (knot construction, bin search,
closed-form forward map / quadratic-root inverse, interval <-> R chart) runs in the 'r' HIP kernel (jf_r_chain_*)."""
import torch
from torch import nn

from . import interval_base
from .. import param_rows
from ... import _hip


class rational_quadratic_spline(interval_base.interval_base):
    def __init__(self, dimension, num_basis_functions=10, euclidean_to_interval_as_first=0, use_permanent_parameters=0, low_boundary=0,
                 high_boundary=1.0, min_width=1e-4, min_height=1e-4, min_derivative=1e-4, fix_boundary_derivatives=-1.0,
                 smooth_second_derivative=0, restrict_max_min_width_height_ratio=-1.0, fix_first_width_n_height_to_zero=0,
                 also_fix_second_width_to_zero=0, independent_width_height_parametrization=0):
        """Symbol "r" (neural spline flows, arXiv:1906.04032).  Parameters as in the reference (:78-94)."""
        super().__init__(dimension=dimension, euclidean_to_interval_as_first=euclidean_to_interval_as_first,
                         use_permanent_parameters=use_permanent_parameters, low_boundary=low_boundary, high_boundary=high_boundary)
        self.num_basis_functions = num_basis_functions
        self.fix_first_width_n_height_to_zero = fix_first_width_n_height_to_zero
        self.also_fix_second_width_to_zero = also_fix_second_width_to_zero
        self.fix_boundary_derivatives = fix_boundary_derivatives
        self.smooth_second_derivative = smooth_second_derivative
        self.min_width, self.min_height, self.min_derivative = min_width, min_height, min_derivative
        self.restrict_max_min_width_height_ratio = restrict_max_min_width_height_ratio
        self.independent_width_height_parametrization = independent_width_height_parametrization
        (self.num_width_params, self.num_height_params, self.num_derivative_params, self._fix_bd,
         self._fix_bd_value) = param_rows.spline_counts(num_basis_functions, fix_first_width_n_height_to_zero, also_fix_second_width_to_zero,
                                                        smooth_second_derivative, fix_boundary_derivatives, min_derivative, circular=False)
        if use_permanent_parameters:
            self.rel_log_widths = nn.Parameter(torch.randn(self.num_width_params).type(torch.double).unsqueeze(0))
            self.rel_log_heights = nn.Parameter(torch.randn(self.num_height_params).type(torch.double).unsqueeze(0))
            if self.num_derivative_params > 0:
                self.rel_log_derivatives = nn.Parameter(torch.randn(self.num_derivative_params).type(torch.double).unsqueeze(0))
        self.total_param_num += self.num_width_params + self.num_height_params + self.num_derivative_params
        self._rows = param_rows.PermanentRowCache()

    def c_struct(self, first=None):
        L = _hip.jf_r_layer()
        L.sp = param_rows.spline_struct(self, self.restrict_max_min_width_height_ratio)
        L.lo, L.hi = float(self.low_boundary), float(self.high_boundary)
        L.first = int(self.euclidean_to_interval_as_first if first is None else first)
        return L

    def _permanent_tensors(self):
        return [self.rel_log_widths, self.rel_log_heights] + ([self.rel_log_derivatives] if self.num_derivative_params > 0 else [])

    def _params_for(self, x, extra_inputs):
        if self.use_permanent_parameters:
            return self._rows.get(self._permanent_tensors(), x, self.total_param_num)
        assert extra_inputs is not None, "Conditional PDF.. require *extra_inputs*"
        assert extra_inputs.shape[0] in (x.shape[0], 1), "Extra inputs must be Tensor of shape B X .. or 1 X .. (broadcasting)"
        return extra_inputs

    def inv_flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False, bins=None, status=None):
        x, log_det = inputs
        return _hip.mchain("r", "inv", x, log_det, self._params_for(x, extra_inputs), [self.c_struct()], 1, bins=bins, status=status)

    def flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False, bins=None, status=None):
        x, log_det = inputs
        return _hip.mchain("r", "fwd", x, log_det, self._params_for(x, extra_inputs), [self.c_struct()], 1, bins=bins, status=status)

    def _get_desired_init_parameters(self):
        n = self.num_width_params + self.num_height_params + self.num_derivative_params
        # 0.54 in log space gives a rather flat spline (:409-413)
        return torch.zeros(n) if self.smooth_second_derivative else torch.ones(n) * 0.54

    def _init_params(self, params):
        c = 0
        self.rel_log_widths.data[0, :] = params[c:c + self.num_width_params]; c += self.num_width_params
        self.rel_log_heights.data[0, :] = params[c:c + self.num_height_params]; c += self.num_height_params
        if self.num_derivative_params > 0:
            self.rel_log_derivatives.data[0, :] = params[c:c + self.num_derivative_params]

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        if self.use_permanent_parameters:
            w, h = self.rel_log_widths, self.rel_log_heights
            d = self.rel_log_derivatives if self.num_derivative_params > 0 else None
        else:
            assert extra_inputs is not None
            w = extra_inputs[:, :self.num_width_params]
            h = extra_inputs[:, self.num_width_params:self.num_width_params + self.num_height_params]
            d = extra_inputs[:, self.num_width_params + self.num_height_params:] if self.num_derivative_params > 0 else None
        param_dict[extra_prefix + "widths"] = w
        param_dict[extra_prefix + "heights"] = h
        if self.smooth_second_derivative == 0:
            param_dict[extra_prefix + "derivatives"] = d
