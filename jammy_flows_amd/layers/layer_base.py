"""Plugin API of a flow layer -- mirrors jammy_flows/layers/layer_base.py:4-100 (same method names, argument meaning)."""
import torch
from torch import nn


class layer_base(nn.Module):
    """Base class of all flow layers.

    Subclasses implement ``flow_mapping`` (sampling direction) and ``inv_flow_mapping`` (log-prob direction); both take
    ``inputs=[x, log_det]`` plus ``extra_inputs`` (``None`` for permanent parameters, else a ``(B | 1, total_param_num)`` row
    block) and return fresh ``(x', log_det')`` tensors -- inputs are never modified (reference contract:
    tests/test_general.py:519, 533-550).
    """

    def __init__(self, dimension=1, always_parametrize_in_embedding_space=0):
        super().__init__()
        self.total_param_num = 0
        self.dimension = dimension
        # manifold layers: is the default coordinate system the embedding space (True) or intrinsic coordinates (False)?
        self.always_parametrize_in_embedding_space = always_parametrize_in_embedding_space

    def get_total_param_num(self):
        return self.total_param_num

    def get_desired_init_parameters(self):
        return torch.randn(self.total_param_num)

    def get_layer_embedded_target_dimension(self):
        return self._embedding_conditional_return_num()

    def get_layer_intrinsic_target_dimension(self):
        return self.dimension

    def get_layer_base_dimension(self):
        return self._get_layer_base_dimension()

    # ---- to be provided by subclasses
    def init_params(self, params):
        raise NotImplementedError

    def flow_mapping(self, input, extra_inputs=None):
        raise NotImplementedError

    def inv_flow_mapping(self, input, extra_inputs=None):
        raise NotImplementedError

    def _embedding_conditional_return(self, x):
        raise NotImplementedError

    def _embedding_conditional_return_num(self):
        raise NotImplementedError

    def transform_target_space(self, x, log_det=0.0, trafo_from="default", trafo_to="embedding"):
        raise NotImplementedError()

    def obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        raise NotImplementedError
