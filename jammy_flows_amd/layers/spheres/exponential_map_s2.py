"""Exponential-map layer 'v' on S2 -- host side (jammy_flows/layers/spheres/exponential_map_s2.py:73-567).
Arithmetic: 'v' HIP kernel (jf_v_chain_f64): gradient-of-potential exponential map, analytic Jacobian, Newton-on-sphere inverse.
float64 only, as in the reference (:450, :493)."""
import torch
from torch import nn

from . import sphere_base
from ... import _hip


class exponential_map_s2(sphere_base.sphere_base):
    FAMILY = "v"

    def __init__(self, dimension, euclidean_to_sphere_as_first=False, use_permanent_parameters=False, exp_map_type="linear", natural_direction=0,
                 num_components=10, add_rotation=0, max_num_newton_iter=1000, mean_parametrization="old"):
        """Symbol "v" (arXiv:0906.0874, arXiv:2002.02428)."""
        super().__init__(dimension=dimension, euclidean_to_sphere_as_first=euclidean_to_sphere_as_first,
                         use_permanent_parameters=use_permanent_parameters, add_rotation=add_rotation)
        if dimension != 2:
            raise Exception("The exponential map flow should be used for dimension 2!")
        if exp_map_type not in _hip.V_KINDS:
            raise NotImplementedError("exp_map_type '%s' has no HIP kernel (linear / quadratic / exponential / splines do)" % exp_map_type)
        if mean_parametrization != "old":
            raise NotImplementedError("mean_parametrization '%s' has no HIP kernel (the reference itself raises a TypeError for it: "
                                      "exponential_map_s2.py:262 calls compute_householder_matrix without hh_iter)" % mean_parametrization)
        self.num_components = num_components
        self.exp_map_type = exp_map_type
        self.natural_direction = natural_direction
        self.max_num_newton_iter = max_num_newton_iter
        self.mean_parametrization = mean_parametrization
        self.num_mu_params = 3
        self.num_potential_pars = self.num_mu_params + {"exponential": 2, "splines": 1 + 3 * 10 + 1}.get(exp_map_type, 1)      # (:124-129)
        if use_permanent_parameters:
            self.potential_pars = nn.Parameter(torch.randn(self.num_potential_pars, self.num_components).unsqueeze(0))
        self.total_param_num += self.num_potential_pars * self.num_components

    def c_struct(self, first):
        L = _hip.jf_v_layer()
        L.num_components = self.num_components
        L.exp_map_type = _hip.V_KINDS[self.exp_map_type]
        L.natural_direction = int(self.natural_direction)
        L.hh_iter = self.num_householder_iter
        L.max_newton_iter = int(self.max_num_newton_iter)
        L.first = int(first)
        return L

    def _fused(self, direction, inputs, extra_inputs, fix_first, **kw):
        assert inputs[0].dtype == torch.float64, "V flow requires float64, otherwise it often will not converge correctly!"
        return super()._fused(direction, inputs, extra_inputs, fix_first, **kw)

    def _layer_tensors(self):
        return [self.potential_pars]

    def _init_params(self, params):
        assert len(params) == self.num_potential_pars * self.num_components
        self.potential_pars.data = params.reshape(1, self.num_potential_pars, self.num_components)

    def _get_desired_init_parameters(self):
        return torch.randn(self.num_potential_pars * self.num_components)

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        param_dict[extra_prefix + "potential_pars"] = (extra_inputs if extra_inputs is not None else self.potential_pars).data
