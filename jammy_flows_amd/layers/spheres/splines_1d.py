"""Circular rational-quadratic spline layer 'o' on S1 -- host side (jammy_flows/layers/spheres/splines_1d.py:8-357).
Arithmetic: 'o' HIP kernel (jf_o_chain_*): periodic / smooth-circular spline + Householder rotation + S1 <-> R chart."""
import torch
from torch import nn

from . import sphere_base
from .. import param_rows
from ... import _hip


class spline_1d(sphere_base.sphere_base):
    FAMILY = "o"

    def __init__(self, dimension=1, euclidean_to_sphere_as_first=True, add_rotation=1, natural_direction=1, use_permanent_parameters=False,
                 num_basis_functions=2, min_width=1e-4, min_height=1e-4, min_derivative=1e-4, fix_boundary_derivatives=-1.0,
                 smooth_second_derivative=0, fix_first_width_n_height_to_zero=0, also_fix_second_width_to_zero=0,
                 independent_width_height_parametrization=0):
        """Symbol "o" (arXiv:2002.02428)."""
        super().__init__(dimension=1, euclidean_to_sphere_as_first=euclidean_to_sphere_as_first, add_rotation=add_rotation,
                         use_permanent_parameters=use_permanent_parameters)
        if dimension != 1:
            raise Exception("The circular spline flow is defined for dimension 1, but dimension %d is handed over" % dimension)
        self.natural_direction = natural_direction
        self.fix_boundary_derivatives = fix_boundary_derivatives
        self.num_basis_functions = num_basis_functions
        self.fix_first_width_n_height_to_zero = fix_first_width_n_height_to_zero
        self.also_fix_second_width_to_zero = also_fix_second_width_to_zero
        self.smooth_second_derivative = smooth_second_derivative
        self.min_width, self.min_height, self.min_derivative = min_width, min_height, min_derivative
        self.independent_width_height_parametrization = independent_width_height_parametrization
        (self.num_width_params, self.num_height_params, self.num_derivative_params, self._fix_bd,
         self._fix_bd_value) = param_rows.spline_counts(num_basis_functions, fix_first_width_n_height_to_zero, also_fix_second_width_to_zero,
                                                        smooth_second_derivative, fix_boundary_derivatives, min_derivative, circular=True)
        if use_permanent_parameters:
            self.rel_log_widths = nn.Parameter(torch.randn(self.num_width_params).type(torch.double).unsqueeze(0))
            self.rel_log_heights = nn.Parameter(torch.randn(self.num_height_params).type(torch.double).unsqueeze(0))
            if self.num_derivative_params > 0:
                self.rel_log_derivatives = nn.Parameter(torch.randn(self.num_derivative_params).type(torch.double).unsqueeze(0))
        self.total_param_num += self.num_width_params + self.num_height_params + self.num_derivative_params

    def c_struct(self, first):
        L = _hip.jf_o_layer()
        L.sp = param_rows.spline_struct(self)
        L.natural_direction = int(self.natural_direction)
        L.hh_iter = self.num_householder_iter
        L.first = int(first)
        return L

    def _layer_tensors(self):
        return [self.rel_log_widths, self.rel_log_heights] + ([self.rel_log_derivatives] if self.num_derivative_params > 0 else [])

    def _get_desired_init_parameters(self):
        n = self.num_width_params + self.num_height_params + self.num_derivative_params
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
            w = extra_inputs[:, :self.num_width_params]
            h = extra_inputs[:, self.num_width_params:self.num_width_params + self.num_height_params]
            d = extra_inputs[:, self.num_width_params + self.num_height_params:] if self.num_derivative_params > 0 else None
        param_dict[extra_prefix + "widths"] = w
        param_dict[extra_prefix + "heights"] = h
        if self.smooth_second_derivative == 0:
            param_dict[extra_prefix + "derivatives"] = d
