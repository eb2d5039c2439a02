"""Base class of the interval layers (plugin API of jammy_flows/layers/intervals/interval_base.py:8-141).

An interval layer lives on [low_boundary, high_boundary]; the first layer of a block additionally carries the chart between the real
line (where the base distribution lives) and the interval.  In-tree layers fuse that chart into their own kernel; for a third-party
subclass that only implements ``_flow_mapping`` / ``_inv_flow_mapping`` it is applied here through the chart kernel (family "c")."""
from ..layer_base import layer_base, flat_coordinates
from ... import _hip


class interval_base(flat_coordinates, layer_base):
    def __init__(self, dimension=1, euclidean_to_interval_as_first=0, use_permanent_parameters=False, low_boundary=0.0, high_boundary=1.0):
        layer_base.__init__(self, dimension=dimension)
        if self.dimension != 1:
            raise AssertionError("we only allow 1-dimensional interval flows")
        if not high_boundary > low_boundary:
            raise AssertionError("empty interval [%s, %s]" % (low_boundary, high_boundary))
        self.low_boundary, self.high_boundary = low_boundary, high_boundary
        self.interval_width = high_boundary - low_boundary
        self.use_permanent_parameters = use_permanent_parameters
        self.euclidean_to_interval_as_first = euclidean_to_interval_as_first

    # ---- chart real line <-> interval (interval_base.py:33-59), on the device
    def _chart(self, direction, inputs):
        c = _hip.jf_c_layer()
        c.kind, c.hh_iter, c.first = 0, 0, 1
        c.lo, c.hi = float(self.low_boundary), float(self.high_boundary)
        return _hip.mchain("c", direction, inputs[0], inputs[1], None, [c], 1)

    def real_line_to_interval(self, inputs):
        return self._chart("fwd", inputs)

    def interval_to_real_line(self, inputs):
        return self._chart("inv", inputs)

    # ---- public mappings for subclasses without a fused kernel (interval_base.py:61-79)
    def flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        start = self.real_line_to_interval(inputs) if self.euclidean_to_interval_as_first else inputs
        return self._flow_mapping(start, extra_inputs=extra_inputs)

    def inv_flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        out = self._inv_flow_mapping(inputs, extra_inputs=extra_inputs)
        return self.interval_to_real_line([out[0], out[1]]) if self.euclidean_to_interval_as_first else (out[0], out[1])

    # ---- bookkeeping
    def init_params(self, params):
        if len(params) != self.total_param_num:
            raise AssertionError("%d initial parameters given, layer has %d" % (len(params), self.total_param_num))
        self._init_params(params)

    def get_desired_init_parameters(self):
        return self._get_desired_init_parameters()

    def obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        self._obtain_layer_param_structure(param_dict, extra_inputs=extra_inputs, previous_x=None, extra_prefix="")
