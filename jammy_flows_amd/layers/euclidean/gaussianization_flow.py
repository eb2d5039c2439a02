"""Gaussianization flow layer 'g' -- host side.

Same constructor arguments, parameter names/shapes (state_dict compatible) and bookkeeping as
jammy_flows/layers/euclidean/gaussianization_flow.py (class gf_block, :50-386, 1116-1222); all arithmetic
(parameter regulation, Householder rotation, logistic-mixture CDF, inverse-CDF stage, bisection + Newton inverse) runs
in the HIP kernels of csrc/gf_kernels.hip through the C ABI (jf_gf_chain_inv_* / jf_gf_chain_fwd_*).
"""
import math

import numpy
import torch
from torch import nn

from . import euclidean_base
from ... import _hip


class gf_block(euclidean_base.euclidean_base):
    def __init__(self,
                 dimension,
                 nonlinear_stretch_type="classic",
                 num_kde=5,
                 num_householder_iter=-1,
                 use_permanent_parameters=False,
                 fit_normalization=0,
                 inverse_function_type="inormal_partly_precise",
                 model_offset=0,
                 softplus_for_width=0,
                 width_smooth_saturation=1,
                 lower_bound_for_widths=0.01,
                 upper_bound_for_widths=100,
                 lower_bound_for_norms=1,
                 upper_bound_for_norms=10,
                 center_mean=0,
                 clamp_widths=0,
                 regulate_normalization=0,
                 add_skewness=0,
                 rotation_mode="householder"):
        """Symbol "g".  Parameters as in the reference (gaussianization_flow.py:71-102)."""
        super().__init__(dimension=dimension, use_permanent_parameters=use_permanent_parameters, model_offset=model_offset)
        if nonlinear_stretch_type not in ("classic", "rq_splines"):
            raise Exception("Unknown non linear stretch type: %s" % nonlinear_stretch_type)
        if inverse_function_type not in _hip.GF_INV_TYPES:
            raise AssertionError("unknown inverse_function_type %s" % inverse_function_type)
        if rotation_mode not in _hip.GF_ROT_MODES and rotation_mode != "none":
            raise Exception("Unknown rotation mode: %s" % rotation_mode)
        if (center_mean or add_skewness) and nonlinear_stretch_type != "classic":
            raise Exception("center_mean / add_skewness belong to the classic (logistic mixture) stretch")
        assert lower_bound_for_widths > 0.0
        if width_smooth_saturation and not softplus_for_width:
            assert upper_bound_for_widths > 0, "We require a maximum saturation level for smooth saturation!"

        self.nonlinear_stretch_type = nonlinear_stretch_type
        self.num_kde = num_kde
        self.fit_normalization = fit_normalization
        self.regulate_normalization = regulate_normalization
        self.inverse_function_type = inverse_function_type
        self.softplus_for_width = softplus_for_width
        self.width_smooth_saturation = width_smooth_saturation
        self.clamp_widths = clamp_widths
        self.width_min = lower_bound_for_widths
        self.width_max = upper_bound_for_widths if upper_bound_for_widths > 0 else None
        self.lower_bound_for_norms = lower_bound_for_norms
        self.upper_bound_for_norms = upper_bound_for_norms
        self.center_mean = center_mean
        self.add_skewness = add_skewness
        self.rotation_mode = rotation_mode

        # Householder rotation (gaussianization_flow.py:172-196)
        self.use_householder = False
        self.householder_iter = 0
        self.num_householder_params = 0
        if rotation_mode == "householder":
            self.householder_iter = dimension if num_householder_iter == -1 else num_householder_iter
            self.use_householder = self.householder_iter > 0
            if self.use_householder:
                self.num_householder_params = self.householder_iter * dimension
                if use_permanent_parameters:
                    self.vs = nn.Parameter(torch.randn(self.householder_iter, dimension).unsqueeze(0))
        self.total_param_num += self.num_householder_params
        # the other rotation parametrisations (:158-170, 198-223); they run in the general-option kernel (csrc/jf_gf_ext.h)
        self.num_triangle_params = self.num_angle_pars = self.num_cayley_pars = 0
        if rotation_mode == "triangular_combination":
            self.num_triangle_params = int(dimension - 1 + dimension * (dimension - 1))
            if use_permanent_parameters and dimension > 1:
                self.triangle_trafo_pars = nn.Parameter(torch.randn(self.num_triangle_params).unsqueeze(0))
        elif rotation_mode == "angles" and dimension > 1:
            self.num_angle_pars = int(dimension * (dimension - 1) / 2)
            if use_permanent_parameters:
                self.angle_pars = nn.Parameter(torch.randn((1, self.num_angle_pars)))
        elif rotation_mode == "cayley" and dimension > 1:
            assert dimension == 2, "Cayley requires 2 dims at the moment"
            self.num_cayley_pars = 1
            if use_permanent_parameters:
                self.cayley_pars = nn.Parameter(torch.randn((1, 1)))
        self.num_rotation_params = self.num_householder_params + self.num_triangle_params + self.num_angle_pars + self.num_cayley_pars
        self.total_param_num += self.num_triangle_params + self.num_angle_pars + self.num_cayley_pars

        self.num_params_datapoints = num_kde * dimension
        self.total_param_num_means = (num_kde - (1 if center_mean else 0)) * dimension
        bandwidth = (4. * numpy.sqrt(math.pi) / ((math.pi ** 4) * num_kde)) ** 0.2      # Gaussianization-flow paper init (:233)
        self.init_log_width = float(numpy.log(bandwidth))
        if nonlinear_stretch_type == "classic":
            if use_permanent_parameters:
                self.kde_means = nn.Parameter(torch.randn(num_kde - (1 if center_mean else 0), dimension).unsqueeze(0))
                self.kde_log_widths = nn.Parameter(torch.ones(num_kde, dimension).unsqueeze(0) * self.init_log_width)
                if fit_normalization:
                    self.kde_log_weights = nn.Parameter(torch.randn(num_kde, dimension).unsqueeze(0))
                if add_skewness:                                  # (:352-368) the first int(K/2) components are skewed to one side, the rest mirrored
                    self.kde_log_skew_exponents = nn.Parameter(torch.randn(num_kde, dimension).unsqueeze(0))
            self.total_param_num += self.total_param_num_means + self.num_params_datapoints * (1 + (1 if fit_normalization else 0) + (1 if add_skewness else 0))
        else:
            # per-dimension rational-quadratic splines with a learnable box and linear tails (:370-382): (D, K) widths / heights,
            # (D, K+1) derivatives, (D, 4) box = (left, ln(width - 0.5), bottom, ln(height - 0.5))
            if num_kde > _hip.JF_SPLINE_MAX_BINS:
                raise NotImplementedError("g layer with rq_splines: at most %d bins per dimension in the HIP kernel" % _hip.JF_SPLINE_MAX_BINS)
            if use_permanent_parameters:
                self.log_widths = nn.Parameter(torch.randn(dimension, num_kde).unsqueeze(0))
                self.log_heights = nn.Parameter(torch.randn(dimension, num_kde).unsqueeze(0))
                self.log_derivatives = nn.Parameter(torch.randn(dimension, num_kde + 1).unsqueeze(0))
                self.boundary_points = nn.Parameter(torch.randn(dimension, 4).unsqueeze(0))
            self.total_param_num += 2 * num_kde * dimension + (num_kde + 1) * dimension + 4 * dimension

        self._row_cache = None
        self._c_struct = None

    # ------------------------------------------------------------------------------------------------------
    def c_struct(self):
        """the jf_gf_layer descriptor of this layer (include/jammy_hip.h)."""
        if self._c_struct is None:
            if self.softplus_for_width:
                mode = _hip.GF_WIDTH_SOFTPLUS
            elif self.width_smooth_saturation:
                mode = _hip.GF_WIDTH_SMOOTH
            else:
                mode = _hip.GF_WIDTH_EXP
            s = _hip.jf_gf_layer()
            s.num_kde = self.num_kde
            s.hh_iter = self.householder_iter if self.use_householder else 0
            s.model_offset = 1 if self.model_offset else 0
            s.fit_normalization = 1 if self.fit_normalization else 0
            s.regulate_normalization = 1 if (self.fit_normalization and self.regulate_normalization) else 0
            s.inverse_function_type = _hip.GF_INV_TYPES[self.inverse_function_type]
            s.width_mode = mode
            s.clamp_widths = 1 if self.clamp_widths else 0
            s.nonlinear_stretch_type = _hip.GF_STRETCH_RQ_SPLINES if self.nonlinear_stretch_type == "rq_splines" else _hip.GF_STRETCH_CLASSIC
            s.width_min = float(self.width_min)
            s.width_max = float(self.width_max) if self.width_max is not None else -1.0
            s.norm_min = float(self.lower_bound_for_norms)
            s.norm_max = float(self.upper_bound_for_norms)
            s.rotation_mode = _hip.GF_ROT_MODES.get(self.rotation_mode, 0)
            s.center_mean = 1 if self.center_mean else 0
            s.add_skewness = 1 if self.add_skewness else 0
            self._c_struct = s
        return self._c_struct

    def _permanent_tensors(self):
        ts = []
        if self.model_offset:
            ts.append(self.offsets)
        if self.use_householder:
            ts.append(self.vs)
        for name, n in (("triangle_trafo_pars", self.num_triangle_params), ("angle_pars", self.num_angle_pars), ("cayley_pars", self.num_cayley_pars)):
            if n > 0:
                ts.append(getattr(self, name))
        if self.nonlinear_stretch_type == "rq_splines":
            return ts + [self.log_widths, self.log_heights, self.log_derivatives, self.boundary_points]
        ts += [self.kde_means, self.kde_log_widths]
        if self.fit_normalization:
            ts.append(self.kde_log_weights)
        if self.add_skewness:
            ts.append(self.kde_log_skew_exponents)
        return ts

    @property
    def has_extended_options(self):
        """options served by the general-option kernel only (no fused block launch, no backward kernel)"""
        return bool(self.center_mean or self.add_skewness or self.rotation_mode not in ("householder", "none"))

    def permanent_row(self, like):
        """the permanent parameters as one (1, total_param_num) row in the extra_inputs layout (cached until a parameter changes)."""
        ts = self._permanent_tensors()
        key = (like.dtype, like.device, tuple((t._version, t.data_ptr()) for t in ts))
        if self._row_cache is None or self._row_cache[0] != key:
            with torch.no_grad():
                row = torch.cat([t.detach().reshape(-1).to(device=like.device, dtype=like.dtype) for t in ts]).reshape(1, -1)
            assert row.shape[1] == self.total_param_num
            self._row_cache = (key, row)
        return self._row_cache[1]

    def _params_for(self, x, extra_inputs):
        if extra_inputs is None:
            if not self.use_permanent_parameters:
                raise ValueError("layer has no permanent parameters: extra_inputs required")
            return self.permanent_row(x)
        if extra_inputs.shape[1] != self.total_param_num:
            raise ValueError("extra_inputs has %d columns, layer needs %d" % (extra_inputs.shape[1], self.total_param_num))
        return extra_inputs

    # ---- plugin API: one fused launch per call (offset + rotation + mixture + inverse-CDF stage)
    def inv_flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False):
        x, log_det = inputs
        return run_chain([self], "inv", x, log_det, self._params_for(x, extra_inputs))

    def flow_mapping(self, inputs, extra_inputs=None, force_embedding_coordinates=False, force_intrinsic_coordinates=False, status=None):
        z, log_det = inputs
        return run_chain([self], "fwd", z, log_det, self._params_for(z, extra_inputs), status=status)

    _inv_flow_mapping = None
    _flow_mapping = None

    # ---- initialisation bookkeeping (gaussianization_flow.py:1116-1222)
    def _get_desired_init_parameters(self):
        vec = []
        if self.num_householder_params > 0:
            vec.append(torch.randn(self.householder_iter * self.dimension))
        vec.append(torch.zeros(self.num_triangle_params + self.num_angle_pars + self.num_cayley_pars))     # identity rotations (:1123-1134)
        if self.nonlinear_stretch_type == "rq_splines":          # (:1150-1166)
            vec.append(torch.ones(self.num_kde * self.dimension))
            vec.append(torch.ones(self.num_kde * self.dimension))
            vec.append(torch.ones((self.num_kde + 1) * self.dimension) * 0.54135)    # softplus^-1(1)
            vec.append(torch.Tensor(self.dimension * [-1.0, 1.0, -1.0, 1.0]))
            return torch.cat(vec)
        vec.append(torch.randn(self.total_param_num_means))
        vec.append(torch.ones(self.num_kde * self.dimension) * self.init_log_width)
        if self.fit_normalization:
            vec.append(torch.ones(self.num_kde * self.dimension))
        if self.add_skewness:
            vec.append(torch.zeros(self.num_kde * self.dimension))
        return torch.cat(vec)

    def _init_params(self, params):
        c = 0
        if self.use_householder:
            self.vs.data = torch.reshape(params[:self.num_householder_params], [1, self.householder_iter, self.dimension])
            c += self.num_householder_params
        for name, m in (("triangle_trafo_pars", self.num_triangle_params), ("angle_pars", self.num_angle_pars), ("cayley_pars", self.num_cayley_pars)):
            if m > 0:          # (:1173-1193; the reference's cayley branch indexes the flat vector as a matrix and cannot run -- the intent is this)
                getattr(self, name).data = torch.reshape(params[c:c + m], [1, m])
                c += m
        n = self.num_params_datapoints
        if self.nonlinear_stretch_type == "rq_splines":          # (:1211-1222)
            K, D = self.num_kde, self.dimension
            self.log_widths.data = torch.reshape(params[c:c + K * D], [1, D, K]); c += K * D
            self.log_heights.data = torch.reshape(params[c:c + K * D], [1, D, K]); c += K * D
            self.log_derivatives.data = torch.reshape(params[c:c + (K + 1) * D], [1, D, K + 1]); c += (K + 1) * D
            self.boundary_points.data = torch.reshape(params[c:c + 4 * D], [1, D, 4])
            return
        nm = self.total_param_num_means
        self.kde_means.data = torch.reshape(params[c:c + nm], [1, self.num_kde - (1 if self.center_mean else 0), self.dimension]); c += nm
        self.kde_log_widths.data = torch.reshape(params[c:c + n], [1, self.num_kde, self.dimension]); c += n
        if self.fit_normalization:
            self.kde_log_weights.data = torch.reshape(params[c:c + n], [1, self.num_kde, self.dimension]); c += n
        if self.add_skewness:
            self.kde_log_skew_exponents.data = torch.reshape(params[c:c + n], [1, self.num_kde, self.dimension]); c += n

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        c = 0
        n = self.num_params_datapoints
        if self.use_householder:
            param_dict[extra_prefix + "vs"] = (self.vs.data.reshape(1, -1) if extra_inputs is None else extra_inputs[:, :self.num_householder_params])
            c += self.num_householder_params
        for key, name, m in (("trianglepars", "triangle_trafo_pars", self.num_triangle_params), ("anglepars", "angle_pars", self.num_angle_pars),
                             ("cayleypars", "cayley_pars", self.num_cayley_pars)):
            if m > 0:
                param_dict[extra_prefix + key] = getattr(self, name).data if extra_inputs is None else extra_inputs[:, c:c + m]
                c += m if extra_inputs is not None else 0
        if self.nonlinear_stretch_type == "rq_splines":
            K, D = self.num_kde, self.dimension
            names = (("log_widths", K), ("log_heights", K), ("log_derivatives", K + 1), ("boundary_points", 4))
            for name, w in names:
                if extra_inputs is None:
                    param_dict[extra_prefix + name] = getattr(self, name).data
                else:
                    param_dict[extra_prefix + name] = extra_inputs[:, c:c + D * w].reshape(-1, D, w)
                    c += D * w
            return
        if extra_inputs is None:
            param_dict[extra_prefix + "means"] = self.kde_means.data
            param_dict[extra_prefix + "log_widths"] = self.kde_log_widths.data
            if self.fit_normalization:
                param_dict[extra_prefix + "log_norms"] = self.kde_log_weights.data
            if self.add_skewness:
                param_dict[extra_prefix + "exponents"] = self.kde_log_skew_exponents.data
        else:
            nm = self.total_param_num_means
            param_dict[extra_prefix + "means"] = extra_inputs[:, c:c + nm].reshape(-1, self.num_kde - (1 if self.center_mean else 0), self.dimension); c += nm
            param_dict[extra_prefix + "log_widths"] = extra_inputs[:, c:c + n].reshape(-1, self.num_kde, self.dimension); c += n
            if self.fit_normalization:
                param_dict[extra_prefix + "log_norms"] = extra_inputs[:, c:c + n].reshape(-1, self.num_kde, self.dimension); c += n
            if self.add_skewness:
                param_dict[extra_prefix + "exponents"] = extra_inputs[:, c:c + n].reshape(-1, self.num_kde, self.dimension)


# ----------------------------------------------------------------------------------------------------------
def chain_fits(layers):
    """does ONE launch of these 'g' layers fit a CU's LDS in every direction the host may ask for (log-prob, sampling, backward; float64,
    broadcast parameters: the most demanding case)?  Wide layers (D > 8: groups of 16 / 32 lanes per row, parameter rows of thousands of
    values) run as several shorter launches.  Cached per layer list (a ctypes query of the library's own bookkeeping)."""
    key = tuple(id(l) for l in layers)
    cache = layers[0].__dict__.setdefault("_chain_fits_cache", {})
    hit = cache.get(key)
    if hit is None:
        try:
            arr = _hip.gf_layer_array([l.c_struct() for l in layers])
            D = layers[0].dimension
            ext = any(l.has_extended_options for l in layers)
            hit = all(_hip.gf_chain_fits(arr, len(layers), D, torch.float64, bcast, backward)
                      for bcast in (True, False) for backward in ((False,) if (ext or any(l.nonlinear_stretch_type != "classic" for l in layers))
                                                                   else (False, True)))
        except _hip.HipUnavailable:
            hit = layers[0].dimension <= 8
        cache[key] = hit
    return hit


def chain_supported(layers):
    """can this list of layers of one e-block be run as one fused launch?"""
    return (1 <= len(layers) <= _hip.JF_MAX_CHAIN and all(type(l) is gf_block for l in layers) and layers[0].dimension <= _hip.GF_MAX_DIM
            and chain_fits(layers))


def chain_permanent_row(layers, like):
    """the permanent rows of a block's layers side by side (cached: rebuilt only when one of the layers' rows was)."""
    rows = [l.permanent_row(like) for l in layers]
    if len(rows) == 1:
        return rows[0]
    key = tuple(id(r) for r in rows)
    hit = getattr(layers[0], "_chain_row_cache", None)
    if hit is None or hit[0] != key:
        hit = (key, torch.cat(rows, dim=1), rows)          # `rows` keeps the ids alive
        layers[0]._chain_row_cache = hit
    return hit[1]


def run_chain(layers, direction, x, log_det, params, x_out=None, base_logp_in=None, want_base_logp=False, status=None, want_total=False):
    """all layers of an e-block in ONE kernel launch.  `params`: (1|B, sum of the layers' total_param_num), layer order 0..n-1
    (what the amortisation MLP emits, main/default.py:1002-1012 / :1488).  want_total: see _hip.gf_chain."""
    arr = _hip.gf_layer_array([l.c_struct() for l in layers])
    return _hip.gf_chain(direction, x, log_det, params, arr, len(layers), layers[0].dimension, x_out=x_out, base_logp_in=base_logp_in,
                         want_base_logp=want_base_logp, status=status, want_total=want_total)
