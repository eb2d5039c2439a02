"""Plugin API of a flow layer (same method names and argument meaning as jammy_flows/layers/layer_base.py:4-100) plus the two
mixins the manifold base classes of this package are assembled from."""
import torch
from torch import nn


def _abstract(name):
    def method(self, *args, **kwargs):
        raise NotImplementedError("%s.%s" % (type(self).__name__, name))
    method.__name__ = name
    return method


class layer_base(nn.Module):
    """A flow layer maps ``[x, log_det]`` in the sampling direction (``flow_mapping``) and in the log-prob direction
    (``inv_flow_mapping``).  ``extra_inputs`` is ``None`` when the layer owns its parameters, else a ``(B | 1, total_param_num)`` row
    block.  Both directions return fresh tensors; the inputs stay untouched (the reference pins that, tests/test_general.py:519, 533-550).
    """

    def __init__(self, dimension=1, always_parametrize_in_embedding_space=0):
        super().__init__()
        self.dimension = dimension
        self.total_param_num = 0
        # manifold layers only: True when the layer's default coordinates are embedding coordinates rather than intrinsic ones
        self.always_parametrize_in_embedding_space = always_parametrize_in_embedding_space

    # sizes
    def get_total_param_num(self):
        return self.total_param_num

    def get_layer_intrinsic_target_dimension(self):
        return self.dimension

    def get_layer_embedded_target_dimension(self):
        return self._embedding_conditional_return_num()

    def get_layer_base_dimension(self):
        return self._get_layer_base_dimension()

    def get_desired_init_parameters(self):
        return torch.randn(self.total_param_num)

    # what a concrete layer (or one of the manifold base classes) has to supply
    init_params = _abstract("init_params")
    flow_mapping = _abstract("flow_mapping")
    inv_flow_mapping = _abstract("inv_flow_mapping")
    transform_target_space = _abstract("transform_target_space")
    obtain_layer_param_structure = _abstract("obtain_layer_param_structure")
    _embedding_conditional_return = _abstract("_embedding_conditional_return")
    _embedding_conditional_return_num = _abstract("_embedding_conditional_return_num")


class flat_coordinates:
    """Mixin for manifolds whose embedding equals their intrinsic coordinates (Euclidean space, intervals): the conditioning of later
    sub-manifolds sees x itself and coordinate transformations are the identity.  Also declares the per-layer hooks."""

    def _embedding_conditional_return(self, x):
        return x

    def _embedding_conditional_return_num(self):
        return self.dimension

    def _get_layer_base_dimension(self):
        return self.dimension

    def transform_target_space(self, x, log_det=0.0, transform_from="default", transform_to="embedding"):
        return x, log_det

    _init_params = _abstract("_init_params")
    _get_desired_init_parameters = _abstract("_get_desired_init_parameters")
    _flow_mapping = _abstract("_flow_mapping")
    _inv_flow_mapping = _abstract("_inv_flow_mapping")
    _obtain_layer_param_structure = _abstract("_obtain_layer_param_structure")


class parameter_free:
    """Mixin of the identity layers ("x", "y", "z"): nothing to learn, nothing to do inside the layer -- whatever their manifold base
    class adds around a layer (offset, rotation, charts) is all that happens."""

    def _flow_mapping(self, inputs, extra_inputs=None, **unused):
        return inputs[0], inputs[1]

    _inv_flow_mapping = _flow_mapping

    def _get_desired_init_parameters(self):
        return torch.Tensor([])

    def _init_params(self, params):
        assert len(params) == 0

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        return None
