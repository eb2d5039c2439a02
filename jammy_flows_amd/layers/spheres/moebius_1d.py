"""Moebius layer 'm' on S1 -- host side (jammy_flows/layers/spheres/moebius_1d.py:11-284).
Arithmetic: 'm' HIP kernel (jf_m_chain_*): convex mixture of Moebius maps, bisection + Newton in the non-natural direction."""
import torch
from torch import nn

from . import sphere_base
from ... import _hip


class moebius(sphere_base.sphere_base):
    FAMILY = "m"

    def __init__(self, dimension=1, euclidean_to_sphere_as_first=True, add_rotation=0, natural_direction=0, use_permanent_parameters=False,
                 use_moebius_xyz_parametrization=True, num_basis_functions=5):
        """Symbol "m" (arXiv:2002.02428)."""
        super().__init__(dimension=1, euclidean_to_sphere_as_first=euclidean_to_sphere_as_first, add_rotation=add_rotation,
                         use_permanent_parameters=use_permanent_parameters)
        if dimension != 1:
            raise Exception("The moebius flow is defined for dimension 1, but dimension %d is handed over" % dimension)
        self.use_moebius_xyz_parametrization = use_moebius_xyz_parametrization
        self.num_basis_functions = num_basis_functions
        self.num_omega_pars = 4 if use_moebius_xyz_parametrization else 3        # (x, y) or the angle of omega, + length + weight (:39-46)
        self.total_param_num += self.num_basis_functions * self.num_omega_pars
        if use_permanent_parameters:
            self.moebius_pars = nn.Parameter(torch.randn(self.num_basis_functions, self.num_omega_pars).type(torch.double).unsqueeze(0))
        self.natural_direction = natural_direction

    def c_struct(self, first):
        L = _hip.jf_m_layer()
        L.num_components = self.num_basis_functions
        L.natural_direction = int(self.natural_direction)
        L.hh_iter = self.num_householder_iter
        L.first = int(first)
        L.omega_pars = self.num_omega_pars
        return L

    def _layer_tensors(self):
        return [self.moebius_pars]

    def _init_params(self, params):
        self.moebius_pars.data = params.reshape(1, self.num_basis_functions, self.num_omega_pars)

    def _get_desired_init_parameters(self):
        return torch.randn(self.num_basis_functions * self.num_omega_pars)

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        pars = self.moebius_pars if extra_inputs is None else extra_inputs.reshape(-1, self.num_basis_functions, self.num_omega_pars)
        param_dict[extra_prefix + "moebius"] = pars.data
