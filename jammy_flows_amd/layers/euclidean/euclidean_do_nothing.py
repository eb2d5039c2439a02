"""Identity layer "x" (reference: jammy_flows/layers/euclidean/euclidean_do_nothing.py); with add_offset=1 the additive offset of
euclidean_base is the whole layer."""
from .euclidean_base import euclidean_base
from ..layer_base import parameter_free


class euclidean_do_nothing(parameter_free, euclidean_base):
    def __init__(self, dimension, use_permanent_parameters=True, add_offset=0):
        euclidean_base.__init__(self, dimension=dimension, use_permanent_parameters=use_permanent_parameters, model_offset=add_offset)
