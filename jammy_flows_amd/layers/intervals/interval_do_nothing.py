"""Identity layer 'z' on an interval (jammy_flows/layers/intervals/interval_do_nothing.py); only the base-class chart acts."""
import torch

from . import interval_base


class interval_do_nothing(interval_base.interval_base):
    def __init__(self, dimension, euclidean_to_interval_as_first=0, use_permanent_parameters=False, low_boundary=0.0, high_boundary=1.0):
        super().__init__(dimension=dimension, euclidean_to_interval_as_first=euclidean_to_interval_as_first,
                         use_permanent_parameters=use_permanent_parameters, low_boundary=low_boundary, high_boundary=high_boundary)

    def _init_params(self, params):
        assert len(params) == 0

    def _get_desired_init_parameters(self):
        return torch.Tensor([])

    def _inv_flow_mapping(self, inputs, extra_inputs=None):
        return inputs[0], inputs[1]

    def _flow_mapping(self, inputs, extra_inputs=None):
        return inputs[0], inputs[1]

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        return
