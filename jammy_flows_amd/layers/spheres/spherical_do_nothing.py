"""Identity layer 'y' on a sphere (jammy_flows/layers/spheres/spherical_do_nothing.py): only the base-class rotation / chart act."""
import torch

from . import sphere_base
from ... import _hip


class spherical_do_nothing(sphere_base.sphere_base):
    FAMILY = "c"

    def __init__(self, dimension, euclidean_to_sphere_as_first=False, use_permanent_parameters=True, add_rotation=0):
        super().__init__(dimension=dimension, euclidean_to_sphere_as_first=euclidean_to_sphere_as_first,
                         use_permanent_parameters=use_permanent_parameters, add_rotation=add_rotation)

    def c_struct(self, first):
        return self._c_struct_base(self.num_householder_iter, first)

    def _init_params(self, params):
        assert len(params) == 0

    def _get_desired_init_parameters(self):
        return torch.Tensor([])

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        return
