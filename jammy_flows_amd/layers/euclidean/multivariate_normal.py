"""'t' -- affine flow / multivariate normal layer: host side of jf_t_layer_* (csrc/t_kernels.hip).

Same constructor arguments, parameter names and parameter-row layout as jammy_flows/layers/euclidean/multivariate_normal.py:54-367
(``single_diagonal_log`` / ``full_diagonal_log`` / ``lower_triangular_entries`` + the ``offsets`` of euclidean_base), so a reference
state_dict loads.  One launch per call: offset, width regulation of the log-diagonal, triangular solve (log-prob direction) or product
(sampling direction) and the log-determinant are fused; the offset is part of the kernel, not a separate torch op."""
import torch
from torch import nn

from . import euclidean_base
from ... import _hip

COV_TYPES = {"identity": 0, "diagonal_symmetric": 1, "diagonal": 2, "full": 3}


class mvn_block(euclidean_base.euclidean_base):
    def __init__(self, dimension, cov_type="full", use_permanent_parameters=False, model_offset=0, width_smooth_saturation=1,
                 lower_bound_for_widths=0.01, upper_bound_for_widths=100, softplus_for_width=0, clamp_widths=0):
        super().__init__(dimension=dimension, use_permanent_parameters=use_permanent_parameters, model_offset=model_offset)
        assert cov_type in COV_TYPES, cov_type
        assert lower_bound_for_widths > 0.0
        if dimension > _hip.T_MAX_DIM:
            raise NotImplementedError("the 't' kernel handles up to %d dimensions" % _hip.T_MAX_DIM)
        self.cov_type = cov_type
        self.width_min = lower_bound_for_widths
        self.width_max = upper_bound_for_widths if upper_bound_for_widths > 0 else None
        self.clamp_widths = clamp_widths
        self.softplus_for_width = softplus_for_width
        self.width_smooth_saturation = width_smooth_saturation
        if width_smooth_saturation:
            assert self.width_max is not None, "We require a maximum saturation level for smooth saturation!"
        n_low = dimension * (dimension - 1) // 2
        if cov_type == "diagonal_symmetric":
            if use_permanent_parameters:
                self.single_diagonal_log = nn.Parameter(torch.randn(1, 1).type(torch.double))
            self.total_param_num += 1
        elif cov_type == "diagonal":
            if use_permanent_parameters:
                self.full_diagonal_log = nn.Parameter(torch.randn(1, dimension).type(torch.double))
            self.total_param_num += dimension
        elif cov_type == "full":
            if use_permanent_parameters:
                self.full_diagonal_log = nn.Parameter(torch.randn(1, dimension).type(torch.double))
                self.lower_triangular_entries = nn.Parameter(torch.randn(1, n_low).type(torch.double))
            self.total_param_num += dimension + n_low
        self._struct = None

    # ------------------------------------------------------------------------------------------------------
    def c_struct(self):
        if self._struct is None:
            s = _hip.jf_t_layer()
            s.cov_type = COV_TYPES[self.cov_type]
            s.model_offset = 1 if self.model_offset else 0
            if self.softplus_for_width:
                s.width_mode = _hip.GF_WIDTH_SOFTPLUS
            elif self.width_smooth_saturation:
                s.width_mode = _hip.GF_WIDTH_SMOOTH
            else:
                s.width_mode = _hip.GF_WIDTH_EXP
            s.clamp_widths = 1 if self.clamp_widths else 0
            s.width_min = float(self.width_min)
            s.width_max = float(self.width_max) if self.width_max is not None else -1.0
            self._struct = s
        return self._struct

    def _permanent_tensors(self):
        ts = [self.offsets] if self.model_offset else []
        if self.cov_type == "diagonal_symmetric":
            ts.append(self.single_diagonal_log)
        elif self.cov_type == "diagonal":
            ts.append(self.full_diagonal_log)
        elif self.cov_type == "full":
            ts += [self.full_diagonal_log, self.lower_triangular_entries]
        return ts

    def _params_for(self, x, extra_inputs):
        if extra_inputs is not None:
            if extra_inputs.shape[1] != self.total_param_num:
                raise ValueError("extra_inputs has %d columns, layer needs %d" % (extra_inputs.shape[1], self.total_param_num))
            return extra_inputs
        if self.total_param_num == 0:
            return None
        assert self.use_permanent_parameters, "layer has no permanent parameters: extra_inputs required"
        with torch.no_grad():
            return torch.cat([t.detach().reshape(-1).to(device=x.device, dtype=x.dtype) for t in self._permanent_tensors()]).reshape(1, -1)

    # ---- plugin API: one launch per call (offset fused)
    def inv_flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False, **kw):
        x, log_det = inputs
        return _hip.t_layer("inv", x, log_det, self._params_for(x, extra_inputs), self.c_struct(), self.dimension, **kw)[:2]

    def flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False, **kw):
        x, log_det = inputs
        return _hip.t_layer("fwd", x, log_det, self._params_for(x, extra_inputs), self.c_struct(), self.dimension, **kw)[:2]

    # ---- bookkeeping
    def _get_desired_init_parameters(self):
        n = {"identity": 0, "diagonal_symmetric": 1, "diagonal": self.dimension,
             "full": self.dimension + self.dimension * (self.dimension - 1) // 2}[self.cov_type]
        return torch.zeros(n)

    def _init_params(self, params):
        D = self.dimension
        if self.cov_type == "diagonal_symmetric":
            self.single_diagonal_log.data = torch.reshape(params[:1], [1, 1]).to(self.single_diagonal_log.dtype)
        elif self.cov_type == "diagonal":
            self.full_diagonal_log.data = torch.reshape(params[:D], [1, D]).to(self.full_diagonal_log.dtype)
        elif self.cov_type == "full":
            self.full_diagonal_log.data = torch.reshape(params[:D], [1, D]).to(self.full_diagonal_log.dtype)
            self.lower_triangular_entries.data = torch.reshape(params[D:], [1, D * (D - 1) // 2]).to(self.lower_triangular_entries.dtype)

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        D = self.dimension
        if self.cov_type == "identity":
            return
        if self.cov_type == "diagonal_symmetric":
            param_dict[extra_prefix + "log_diagonal_symmetric"] = (extra_inputs if extra_inputs is not None else self.single_diagonal_log).data
        else:
            param_dict[extra_prefix + "log_diagonal"] = (extra_inputs[:, :D] if extra_inputs is not None else self.full_diagonal_log).data
            if self.cov_type == "full":
                param_dict[extra_prefix + "lower_trinagular_entries"] = (extra_inputs[:, D:] if extra_inputs is not None
                                                                         else self.lower_triangular_entries).data
