"""Identity layer "z" on an interval (reference: jammy_flows/layers/intervals/interval_do_nothing.py): only the chart of
interval_base between the interval and the real line acts, and only when the layer is the first of its block."""
from .interval_base import interval_base
from ..layer_base import parameter_free


class interval_do_nothing(parameter_free, interval_base):
    def __init__(self, dimension, euclidean_to_interval_as_first=0, use_permanent_parameters=False, low_boundary=0.0, high_boundary=1.0):
        interval_base.__init__(self, dimension=dimension, euclidean_to_interval_as_first=euclidean_to_interval_as_first,
                               use_permanent_parameters=use_permanent_parameters, low_boundary=low_boundary, high_boundary=high_boundary)
