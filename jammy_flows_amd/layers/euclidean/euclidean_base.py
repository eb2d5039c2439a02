"""Base class of Euclidean flow layers -- API of jammy_flows/layers/euclidean/euclidean_base.py:8-160.

The base class only knows about the optional additive offset of the last layer of an e-block.  For third-party subclasses that
implement ``_flow_mapping`` / ``_inv_flow_mapping`` with their own code the offset is applied here; the in-tree layer
(``gf_block``) overrides both public methods and fuses the offset into its HIP kernel.
"""
import torch
from torch import nn

from .. import layer_base
from ... import _hip


class euclidean_base(layer_base.layer_base):
    def __init__(self, dimension=1, use_permanent_parameters=False, model_offset=0):
        super().__init__(dimension=dimension)
        self.use_permanent_parameters = use_permanent_parameters
        self.model_offset = model_offset
        self.offsets = None
        if self.model_offset:
            if self.use_permanent_parameters:
                # created as double like the reference does (euclidean_base.py:26-29); shape becomes (D,) after init_params
                self.offsets = nn.Parameter(torch.randn(dimension).type(torch.double).unsqueeze(0))
            self.total_param_num += dimension

    # ---- public plugin API
    def _offset_and_rest(self, x, extra_inputs):
        off = None
        if self.offsets is not None:
            off = self.offsets.to(x).reshape(1, -1)
        rest = extra_inputs
        if extra_inputs is not None:
            eo = extra_inputs[:, :self.dimension]
            off = eo if off is None else off + eo
            rest = extra_inputs[:, self.dimension:]
        return off, rest

    def inv_flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        if not self.model_offset:
            return self._inv_flow_mapping(inputs, extra_inputs=extra_inputs)
        x, log_det = inputs
        _hip.require_device(x)
        off, rest = self._offset_and_rest(x, extra_inputs)
        return self._inv_flow_mapping([x - off, log_det], extra_inputs=rest)

    def flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        if not self.model_offset:
            return self._flow_mapping(inputs, extra_inputs=extra_inputs)
        x, log_det = inputs
        _hip.require_device(x)
        off, rest = self._offset_and_rest(x, extra_inputs)
        y, log_det = self._flow_mapping([x, log_det], extra_inputs=rest)
        return [y + off, log_det]

    # ---- bookkeeping used by the pdf orchestrator
    def get_desired_init_parameters(self):
        parts = []
        if self.model_offset:
            parts.append(torch.ones(self.dimension) * 0.001)
        parts.append(self._get_desired_init_parameters())
        return torch.cat(parts)

    def init_params(self, params):
        assert len(params) == self.total_param_num, (len(params), self.total_param_num)
        assert self.use_permanent_parameters == 1, "init_params is only defined for layers with permanent parameters"
        if self.model_offset:
            self.offsets.data = params[:self.dimension].to(self.offsets.data.dtype)
            self._init_params(params[self.dimension:])
        else:
            self._init_params(params)

    def _embedding_conditional_return(self, x):
        return x

    def _embedding_conditional_return_num(self):
        return self.dimension

    def _get_layer_base_dimension(self):
        return self.dimension

    def transform_target_space(self, x, log_det=0.0, transform_from="default", transform_to="embedding"):
        return x, log_det

    # ---- provided by concrete layers
    def _init_params(self, params):
        raise NotImplementedError

    def _get_desired_init_parameters(self):
        raise NotImplementedError

    def _inv_flow_mapping(self, inputs, extra_inputs=None):
        raise NotImplementedError

    def _flow_mapping(self, inputs, extra_inputs=None):
        raise NotImplementedError

    def obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        if self.model_offset:
            if extra_inputs is not None:
                param_dict["offset"] = extra_inputs[:, :self.dimension]
                extra_inputs = extra_inputs[:, self.dimension:]
            else:
                param_dict["offset"] = self.offsets.data
        self._obtain_layer_param_structure(param_dict, extra_inputs=extra_inputs, previous_x=previous_x, extra_prefix=extra_prefix)

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        raise NotImplementedError
