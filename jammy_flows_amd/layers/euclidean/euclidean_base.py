"""Base class of the Euclidean layers (plugin API of jammy_flows/layers/euclidean/euclidean_base.py:8-160).

All it adds to a layer is the optional additive offset of the LAST layer of an e-block (``model_offset``), whose D values sit in front
of the layer's own parameters (euclidean_base.py:36-45, 63-68).  ``gf_block`` overrides the public mappings and fuses the offset into its
kernel; the code here serves subclasses that only bring ``_flow_mapping`` / ``_inv_flow_mapping``."""
import torch
from torch import nn

from ..layer_base import layer_base, flat_coordinates
from ... import _hip


class euclidean_base(flat_coordinates, layer_base):
    def __init__(self, dimension=1, use_permanent_parameters=False, model_offset=0):
        layer_base.__init__(self, dimension=dimension)
        self.use_permanent_parameters = use_permanent_parameters
        self.model_offset = model_offset
        self.offsets = None
        if model_offset:
            self.total_param_num += dimension
            if use_permanent_parameters:
                # a double parameter like the reference's (euclidean_base.py:26-29); init_params later reshapes it to (D,)
                self.offsets = nn.Parameter(torch.randn(dimension).type(torch.double).unsqueeze(0))

    def _split_offset(self, x, extra_inputs):
        """-> (offset row block broadcastable to x, the layer's own parameter block)"""
        shift = None if self.offsets is None else self.offsets.to(x).reshape(1, -1)
        if extra_inputs is None:
            return shift, None
        head, tail = extra_inputs[:, :self.dimension], extra_inputs[:, self.dimension:]
        return (head if shift is None else shift + head), tail

    def flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        if not self.model_offset:
            return self._flow_mapping(inputs, extra_inputs=extra_inputs)
        _hip.require_device(inputs[0])
        shift, own = self._split_offset(inputs[0], extra_inputs)
        y, log_det = self._flow_mapping([inputs[0], inputs[1]], extra_inputs=own)
        return [y + shift, log_det]

    def inv_flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        if not self.model_offset:
            return self._inv_flow_mapping(inputs, extra_inputs=extra_inputs)
        _hip.require_device(inputs[0])
        shift, own = self._split_offset(inputs[0], extra_inputs)
        return self._inv_flow_mapping([inputs[0] - shift, inputs[1]], extra_inputs=own)

    # ---- bookkeeping for the pdf orchestrator
    def get_desired_init_parameters(self):
        own = self._get_desired_init_parameters()
        return torch.cat([torch.full((self.dimension,), 0.001), own]) if self.model_offset else own

    def init_params(self, params):
        if len(params) != self.total_param_num:
            raise AssertionError("%d initial parameters given, layer has %d" % (len(params), self.total_param_num))
        assert self.use_permanent_parameters == 1, "init_params is only defined for layers with permanent parameters"
        own = params
        if self.model_offset:
            self.offsets.data = params[:self.dimension].to(self.offsets.data.dtype)
            own = params[self.dimension:]
        self._init_params(own)

    def obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        if self.model_offset:
            param_dict["offset"] = self.offsets.data if extra_inputs is None else extra_inputs[:, :self.dimension]
            if extra_inputs is not None:
                extra_inputs = extra_inputs[:, self.dimension:]
        self._obtain_layer_param_structure(param_dict, extra_inputs=extra_inputs, previous_x=previous_x, extra_prefix=extra_prefix)
