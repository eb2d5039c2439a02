"""Base class of interval flow layers -- API of jammy_flows/layers/intervals/interval_base.py:8-141.
The R <-> [a,b] chart of the first layer runs in the 'c' (chart) HIP kernel; in-tree layers fuse it into their own kernel."""
import torch

from .. import layer_base
from ... import _hip


class interval_base(layer_base.layer_base):
    def __init__(self, dimension=1, euclidean_to_interval_as_first=0, use_permanent_parameters=False, low_boundary=0.0, high_boundary=1.0):
        super().__init__(dimension=dimension)
        assert self.dimension == 1, "we only allow 1-dimensional interval flows"
        self.use_permanent_parameters = use_permanent_parameters
        self.low_boundary = low_boundary
        self.high_boundary = high_boundary
        self.interval_width = high_boundary - low_boundary
        self.euclidean_to_interval_as_first = euclidean_to_interval_as_first
        assert self.high_boundary > self.low_boundary

    def _chart_struct(self):
        c = _hip.jf_c_layer()
        c.kind, c.hh_iter, c.first = 0, 0, 1
        c.lo, c.hi = float(self.low_boundary), float(self.high_boundary)
        return c

    def real_line_to_interval(self, inputs):
        x, log_det = inputs
        return _hip.mchain("c", "fwd", x, log_det, None, [self._chart_struct()], 1)

    def interval_to_real_line(self, inputs):
        x, log_det = inputs
        return _hip.mchain("c", "inv", x, log_det, None, [self._chart_struct()], 1)

    # generic path for third-party subclasses (interval_base.py:61-79)
    def inv_flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        res, log_det = self._inv_flow_mapping(inputs, extra_inputs=extra_inputs)
        if self.euclidean_to_interval_as_first:
            res, log_det = self.interval_to_real_line([res, log_det])
        return res, log_det

    def flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        if self.euclidean_to_interval_as_first:
            inputs = self.real_line_to_interval(inputs)
        return self._flow_mapping(inputs, extra_inputs=extra_inputs)

    def get_desired_init_parameters(self):
        return self._get_desired_init_parameters()

    def init_params(self, params):
        assert len(params) == self.total_param_num
        self._init_params(params)

    def _embedding_conditional_return(self, x):
        return x

    def _embedding_conditional_return_num(self):
        return self.dimension

    def _get_layer_base_dimension(self):
        return self.dimension

    def transform_target_space(self, x, log_det=0.0, transform_from="default", transform_to="embedding"):
        return x, log_det

    def _init_params(self, params):
        raise NotImplementedError

    def _get_desired_init_parameters(self):
        raise NotImplementedError

    def _inv_flow_mapping(self, inputs, extra_inputs=None):
        raise NotImplementedError

    def _flow_mapping(self, inputs, extra_inputs=None):
        raise NotImplementedError

    def obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        self._obtain_layer_param_structure(param_dict, extra_inputs=extra_inputs, previous_x=None, extra_prefix="")

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        raise NotImplementedError
