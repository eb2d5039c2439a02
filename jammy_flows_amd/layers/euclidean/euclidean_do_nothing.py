"""Identity layer 'x' (jammy_flows/layers/euclidean/euclidean_do_nothing.py)."""
import torch

from . import euclidean_base


class euclidean_do_nothing(euclidean_base.euclidean_base):
    def __init__(self, dimension, use_permanent_parameters=True, add_offset=0):
        super().__init__(dimension=dimension, use_permanent_parameters=use_permanent_parameters, model_offset=add_offset)

    def _flow_mapping(self, inputs, extra_inputs=None):
        z, log_det = inputs
        return z, log_det

    def _inv_flow_mapping(self, inputs, extra_inputs=None):
        x, log_det = inputs
        return x, log_det

    def _init_params(self, params):
        assert len(params) == 0

    def _get_desired_init_parameters(self):
        return torch.Tensor([])

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        return
