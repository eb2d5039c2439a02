"""Autoregressive orchestration of the reference (jammy_flows/main/default.py: class pdf) restated in numpy.
Oracle = test infrastructure only.

  construction rules      main/default.py:153-325 (options), :378-479 (layers), :571-722 (MLPs)
  log-prob                :879-1057 (all_layer_inverse), :1059-1117 (forward)
  sampling                :1373-1531 (all_layer_forward), :1533-1707 (_obtain_sample with injected base noise)
"""
import copy

import numpy as np

from . import gf, mvn
from .mlp import AmortizableMLP, AmortizableMLPSpec, SequentialMLP, list_from_str
from .s2_layers import FLayer, VLayer
from .sphere_layers import MLayer, OLayer, RLayer
from .special import normal_logpdf_sum

# default option values of the in-scope layers (flow_options.py:32-240) -- data, the user-visible API
DEFAULTS = {
    "g": dict(fit_normalization=1, num_householder_iter=-1, num_kde=10, inverse_function_type="isigmoid",
              replace_first_sigmoid_with_icdf=1, skip_model_offset=0, softplus_for_width=0, upper_bound_for_widths=100,
              lower_bound_for_widths=0.01, upper_bound_for_norms=10, lower_bound_for_norms=1, center_mean=0, clamp_widths=0,
              width_smooth_saturation=1, regulate_normalization=1, add_skewness=0, rotation_mode="householder",
              nonlinear_stretch_type="classic"),
    "t": dict(skip_model_offset=0, softplus_for_width=0, upper_bound_for_widths=100, lower_bound_for_widths=0.01, clamp_widths=0,
              width_smooth_saturation=1, cov_type="diagonal"),
    "m": dict(add_rotation=0, num_basis_functions=5, natural_direction=0),
    "o": dict(add_rotation=1, num_basis_functions=2, natural_direction=1, fix_boundary_derivatives=-1.0,
              smooth_second_derivative=1, fix_first_width_n_height_to_zero=0, also_fix_second_width_to_zero=0,
              independent_width_height_parametrization=0, min_width=1e-4, min_height=1e-4, min_derivative=1e-4),
    "v": dict(exp_map_type="exponential", num_components=10, natural_direction=0, add_rotation=0, max_num_newton_iter=1000,
              mean_parametrization="old"),
    "f": dict(add_vertical_rq_spline_flow=0, add_circular_rq_spline_flow=0, add_correlated_rq_spline_flow=0,
              circular_flow_defs="oo", vertical_flow_defs="rr", correlated_max_rank=3, inverse_z_scaling=1,
              boundary_cos_theta_identity_region=0.0, spline_num_basis_functions=5, vertical_smooth=0,
              vertical_restrict_max_min_width_height_ratio=-1.0, vertical_fix_boundary_derivative=1,
              vertical_fix_first_width_n_height_to_zero=0, vertical_also_fix_second_width_to_zero=0,
              vertical_independent_width_height_parametrization=0, circular_add_rotation=0, min_kappa=1e-10,
              kappa_prediction="direct_log_real_bounded", add_extra_rotation_inbetween=0, add_rotation=1,
              rotation_mode="householder", kappa_clamping=0, num_householder_iter=-1),
    "r": dict(num_basis_functions=5, fix_boundary_derivatives=-1.0, smooth_second_derivative=0,
              restrict_max_min_width_height_ratio=-1.0, fix_first_width_n_height_to_zero=0, also_fix_second_width_to_zero=0,
              independent_width_height_parametrization=0, min_width=1e-4, min_height=1e-4, min_derivative=1e-4),
    "x": dict(add_offset=0), "y": dict(add_rotation=0), "z": dict(),
}
LAYER_KIND = {"g": "e", "t": "e", "x": "e", "m": "s", "o": "s", "v": "s", "f": "s", "y": "s", "r": "i", "z": "i"}


class _Identity:
    total_param_num = 0

    def __init__(self, embed_fn=None):
        self._embed = embed_fn

    def row_from_state(self, sd, prefix):
        return np.zeros((1, 0))

    def inverse(self, x, log_det, params, **kw):
        return x, log_det, []

    forward = inverse


class _GLayer:
    def __init__(self, dim, opts, model_offset):
        self.spec = gf.GfSpec(dim, opts, model_offset)
        self.total_param_num = self.spec.total_param_num

    def row_from_state(self, sd, prefix):
        return self.spec.row_from_state(sd, prefix)

    def inverse(self, x, log_det, params):
        return gf.inverse(self.spec, x, log_det, params)

    def forward(self, x, log_det, params):
        return gf.forward(self.spec, x, log_det, params)


class _TLayer:
    def __init__(self, dim, opts, model_offset):
        self.spec = mvn.TSpec(dim, opts, model_offset)
        self.total_param_num = self.spec.total_param_num

    def row_from_state(self, sd, prefix):
        return self.spec.row_from_state(sd, prefix)

    def inverse(self, x, log_det, params):
        return mvn.inverse(self.spec, x, log_det, params)

    def forward(self, x, log_det, params):
        return mvn.forward(self.spec, x, log_det, params)


def _resolve_options(letter, sub_index, layer_index, overwrite):
    """options_overwrite precedence (main/default.py:193-272): (sub,layer) tuple > sub index > letter."""
    if letter == "n":   # legacy name of the S2 autoregressive spline layer
        letter = "f"
    opts = copy.deepcopy(DEFAULTS[letter])
    found = False
    for k, v in overwrite.items():
        if isinstance(k, tuple) and k == (sub_index, layer_index):
            found = True
            opts.update(v[letter])
    if not found:
        for k, v in overwrite.items():
            if isinstance(k, int) and not isinstance(k, bool) and k == sub_index and letter in v:
                found = True
                opts.update(v[letter])
    if not found and letter in overwrite:
        opts.update(overwrite[letter])
    return letter, opts


class OraclePdf:
    def __init__(self, pdf_defs, flow_defs, options_overwrite=None, conditional_input_dim=None, amortization_mlp_dims="128",
                 amortization_mlp_use_custom_mode=False, amortization_mlp_ranks=0, amortize_everything=False,
                 use_as_passthrough_instead_of_pdf=False, state_dict=None, amortization_mlp_highway_mode=0, **unused):
        overwrite = options_overwrite or {}
        self.pdf_defs = pdf_defs.split("+")
        self.flow_defs = flow_defs.split("+")
        assert len(self.pdf_defs) == len(self.flow_defs)
        self.cdim = conditional_input_dim
        self.amortize_everything = amortize_everything
        self.passthrough = use_as_passthrough_instead_of_pdf
        nsub = len(self.pdf_defs)
        mlp_dims = [amortization_mlp_dims] * nsub if isinstance(amortization_mlp_dims, str) else amortization_mlp_dims
        mlp_ranks = [amortization_mlp_ranks] * nsub if isinstance(amortization_mlp_ranks, (int, str)) else amortization_mlp_ranks
        permanent_first = (conditional_input_dim is None) and (not amortize_everything)

        def nested_factory(pd, fd, ow, mlp_dims="128", mlp_ranks=0):
            return OraclePdf(pd, fd, options_overwrite=ow, amortize_everything=True, amortization_mlp_use_custom_mode=True,
                             use_as_passthrough_instead_of_pdf=True, amortization_mlp_dims=mlp_dims,
                             amortization_mlp_ranks=mlp_ranks)

        # ---- layers (init_flow_structure, main/default.py:378-479)
        self.blocks = []
        for si, (sub, letters) in enumerate(zip(self.pdf_defs, self.flow_defs)):
            kind = sub[0]
            dim = int(sub.split("_")[0][1:])
            layers = []
            for li, letter in enumerate(letters):
                letter, o = _resolve_options(letter, si, li, overwrite)
                assert LAYER_KIND[letter] == kind, (letter, sub)
                first = (li == 0) and not self.passthrough
                if kind == "e":
                    if letter == "x":
                        layers.append(_Identity())
                        continue
                    model_offset = 0
                    if li == len(letters) - 1 and o["skip_model_offset"] == 0:
                        model_offset = 1
                    elif li == 0 and letter == "g" and o["replace_first_sigmoid_with_icdf"] > 0 and o["inverse_function_type"] == "isigmoid":
                        o["inverse_function_type"] = "inormal_partly_precise"
                    layers.append(_GLayer(dim, o, model_offset) if letter == "g" else _TLayer(dim, o, model_offset))
                elif kind == "i":
                    parts = sub.split("_")
                    lo, hi = (0.0, 1.0) if len(parts) == 1 else (float(parts[1]), float(parts[2]))
                    if letter == "z":
                        raise NotImplementedError("oracle: z")
                    layers.append(RLayer(dim, o, first, lo, hi))
                else:
                    if letter == "o":
                        layers.append(OLayer(dim, o, first, False))
                    elif letter == "m":
                        layers.append(MLayer(dim, o, first, False))
                    elif letter == "f":
                        layers.append(FLayer(dim, o, first, False, nested_factory))
                    elif letter == "v":
                        layers.append(VLayer(dim, o, first, False))
                    else:
                        raise NotImplementedError("oracle: %s" % letter)
            self.blocks.append(dict(kind=kind, dim=dim, layers=layers, permanent=permanent_first and si == 0,
                                    nparams=sum(l.total_param_num for l in layers)))
        self.total_base_dim = sum(b["dim"] for b in self.blocks)
        self.total_target_dim = self.total_base_dim

        # ---- MLPs (init_encoding_structure, main/default.py:571-722)
        self.mlp_specs = []
        self.total_number_amortizable_params = 0 if amortize_everything else None
        prev_embed = 0
        for si, blk in enumerate(self.blocks):
            emb = blk["dim"] if blk["kind"] in "ei" else blk["dim"] + 1
            spec = None
            if si == 0 and conditional_input_dim is None:
                if amortize_everything:
                    self.total_number_amortizable_params += blk["nparams"]
            elif blk["nparams"] > 0:
                in_dim = prev_embed + (conditional_input_dim or 0)
                if amortization_mlp_use_custom_mode:
                    spec = ("custom", AmortizableMLPSpec(in_dim, mlp_dims[si], blk["nparams"], mlp_ranks[si], amortization_mlp_highway_mode))
                    if amortize_everything:
                        self.total_number_amortizable_params += spec[1].num_amortization_params
                else:
                    spec = ("sequential", len(list_from_str(mlp_dims[si])) + 1)
            self.mlp_specs.append(spec)
            prev_embed += emb
        self.mlps = [None] * nsub
        self.rows = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ------------------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        self.rows = []
        for si, blk in enumerate(self.blocks):
            if blk["permanent"]:
                self.rows.append([l.row_from_state(sd, "layer_list.%d.%d." % (si, li)) for li, l in enumerate(blk["layers"])])
            else:
                self.rows.append(None)
            spec = self.mlp_specs[si]
            if spec is None:
                self.mlps[si] = None
            elif spec[0] == "custom":
                uvb = None if self.amortize_everything else sd["mlp_predictors.%d.u_v_b_pars" % si]
                self.mlps[si] = AmortizableMLP(spec[1], uvb)
            else:
                self.mlps[si] = SequentialMLP(sd, "mlp_predictors.%d." % si, spec[1])

    def _ensure_mlps(self):
        for si, spec in enumerate(self.mlp_specs):
            if spec is not None and self.mlps[si] is None:
                assert spec[0] == "custom" and self.amortize_everything
                self.mlps[si] = AmortizableMLP(spec[1], None)

    def _embed(self, blk, x):
        if blk["kind"] == "s":
            return blk["layers"][-1].embed(x)
        return x

    def _block_params(self, si, cond, embeds, amort, counter):
        """per-layer parameter row blocks of sub-pdf si (main/default.py:936-993 / 1420-1475)."""
        blk = self.blocks[si]
        if blk["permanent"]:
            return self.rows[si], counter
        mlp = self.mlps[si]
        if mlp is not None:
            inp = ([cond] if cond is not None else []) + embeds
            inp = np.concatenate(inp, axis=1)
            if amort is not None:
                n = mlp.num_amortization_params
                full = mlp(inp, extra_inputs=amort[:, counter:counter + n])
                counter += n
            else:
                full = mlp(inp)
        elif self.amortize_everything and blk["nparams"] > 0:
            full = amort[:, counter:counter + blk["nparams"]]
            counter += blk["nparams"]
        else:
            full = np.zeros((1, 0))
        rows, c = [], 0
        for l in blk["layers"]:
            rows.append(full[:, c:c + l.total_param_num])
            c += l.total_param_num
        return rows, counter

    def _to_default(self, x, log_det, from_embedding):
        """transform_target_space embedding->default for intrinsic-by-default layers (main/default.py:1737-1813)."""
        if not from_embedding:
            return x, log_det
        from . import manifolds as mf
        cols, c = [], 0
        for blk in self.blocks:
            if blk["kind"] == "s":
                xe = x[:, c:c + blk["dim"] + 1]
                c += blk["dim"] + 1
                xi, log_det = mf.eucl_to_spherical(xe, log_det, blk["dim"])
                cols.append(xi)
            else:
                cols.append(x[:, c:c + blk["dim"]])
                c += blk["dim"]
        return np.concatenate(cols, axis=1), log_det

    def _to_embedding(self, x, log_det):
        from . import manifolds as mf
        cols, c = [], 0
        for blk in self.blocks:
            xi = x[:, c:c + blk["dim"]]
            c += blk["dim"]
            if blk["kind"] == "s":
                xi, log_det = mf.spherical_to_eucl(xi, log_det, blk["dim"])
            cols.append(xi)
        return np.concatenate(cols, axis=1), log_det

    # ------------------------------------------------------------------------------------------
    def all_layer_inverse(self, x, log_det, cond=None, amortization_parameters=None, trace=None):
        self._ensure_mlps()
        embeds, bases, bins, counter, c = [], [], [], 0, 0
        for si, blk in enumerate(self.blocks):
            rows, counter = self._block_params(si, cond, embeds, amortization_parameters, counter)
            tgt = x[:, c:c + blk["dim"]]
            cur = tgt
            for li in reversed(range(len(blk["layers"]))):
                cur, log_det, b = blk["layers"][li].inverse(cur, log_det, rows[li])
                bins += b
                if trace is not None:
                    trace.append(("%d.%d" % (si, li), cur.copy(), log_det.copy()))
            bases.append(cur)
            embeds.append(self._embed(blk, tgt))
            c += blk["dim"]
        return np.concatenate(bases, axis=1), log_det, bins

    def all_layer_forward(self, z, log_det, cond=None, amortization_parameters=None, trace=None):
        self._ensure_mlps()
        embeds, outs, bins, counter, c = [], [], [], 0, 0
        for si, blk in enumerate(self.blocks):
            rows, counter = self._block_params(si, cond, embeds, amortization_parameters, counter)
            cur = z[:, c:c + blk["dim"]]
            for li in range(len(blk["layers"])):
                cur, log_det, b = blk["layers"][li].forward(cur, log_det, rows[li])
                bins += b
                if trace is not None:
                    trace.append(("%d.%d" % (si, li), cur.copy(), log_det.copy()))
            outs.append(cur)
            embeds.append(self._embed(blk, cur))
            c += blk["dim"]
        return np.concatenate(outs, axis=1), log_det, bins

    def forward(self, x, cond=None, force_embedding_coordinates=False, trace=None, return_bins=False, amortization_parameters=None):
        """-> (log_prob, log_prob_base, base_pos)  (main/default.py:1059-1117)."""
        assert not self.passthrough
        x = np.asarray(x, dtype=np.float64)
        log_det = np.zeros(x.shape[0])
        x, log_det = self._to_default(x, log_det, force_embedding_coordinates)
        base, log_det, bins = self.all_layer_inverse(x, log_det, cond, amortization_parameters=amortization_parameters, trace=trace)
        lp = normal_logpdf_sum(base)
        if return_bins:
            return lp + log_det, lp, base, bins
        return lp + log_det, lp, base

    def sample_from_base(self, z, cond=None, force_embedding_coordinates=False, trace=None, return_bins=False, amortization_parameters=None):
        """_obtain_sample(predefined_target_input=z) -> (x, log_prob, log_prob_base)  (main/default.py:1634-1707)."""
        assert not self.passthrough
        z = np.asarray(z, dtype=np.float64)
        lg = normal_logpdf_sum(z)
        x, log_det, bins = self.all_layer_forward(z, np.zeros(z.shape[0]), cond, amortization_parameters=amortization_parameters, trace=trace)
        if force_embedding_coordinates:
            x, log_det = self._to_embedding(x, log_det)
        if return_bins:
            return x, -log_det + lg, lg, bins
        return x, -log_det + lg, lg
