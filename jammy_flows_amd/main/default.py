"""`pdf` -- the autoregressive orchestrator over a product of manifolds, host side of the MI355X hot path.

User-facing API of jammy_flows/main/default.py (class pdf): same constructor arguments, model strings ("e4+s2+e4" / "gggg+f+gggg"),
``forward`` -> (log_prob, log_prob_base, base_pos), ``sample`` -> (x, base, log_prob, log_prob_base), ``all_layer_inverse`` /
``all_layer_forward``, ``transform_target_space``, ``init_params``, ``count_parameters`` and state_dict-compatible modules.
What differs is underneath: every sub-pdf is evaluated by fused HIP kernels (one launch per Euclidean block, one per manifold
layer, MFMA launches for the amortisation MLPs) instead of thousands of eager ops.

Reference line numbers cited below refer to jammy_flows/main/default.py.
"""
import copy
import os

import numpy
import torch
import torch.utils._python_dispatch
import torch.utils._pytree
from torch import nn

from .. import _hip, autograd, init_fns
from ..amortizable_mlp import AmortizableMLP
from ..extra_functions import list_from_str
from ..flow_options import canonical, check_flow_option, layer_class, obtain_default_options, opts_dict
from ..layers.euclidean import gaussianization_flow as gfl


# float64 amortisation MLPs with a wide output: "i8x6" / "i8x5" = second product as int8 digit-slice products (operands kept to 2^-41 / 2^-34,
# csrc/mlp_i8_kernels.hip), "f64" = jf_mlp2_f64 on the float64 matrix cores.  A one-element list so that tests and benchmarks can switch it.
# Default i8x5 since round 5: its parameters agree with the float64 product to 2e-11 relative (log p of C3 to 2e-8 either way, bar 1e-4, the
# parity tests' 1e-7) and the 128 -> 548 product needs 15 slice-pair passes instead of 21 (-0.2 ms per 2^20 rows).
MLP_MATRIX_ARITHMETIC_F64 = [os.environ.get("JF_MLP_MATRIX_ARITHMETIC_F64", "i8x5")]
MLP_I8_MIN_ROWS, MLP_I8_MIN_COLS = [4096], [128]         # below these the exact kernel costs microseconds and the digit image is not worth building
if MLP_MATRIX_ARITHMETIC_F64[0] not in ("i8x6", "i8x5", "f64"):
    raise ValueError("JF_MLP_MATRIX_ARITHMETIC_F64 must be 'i8x6', 'i8x5' or 'f64', got %r" % MLP_MATRIX_ARITHMETIC_F64[0])


class HipLinearStack(nn.Sequential):
    """nn.Sequential(Linear, Tanh, ..., Linear) of the default amortisation MLP (:656-670) -- identical module names, so the
    reference's ``mlp_predictors.N.{0,2,..}.{weight,bias}`` state loads -- evaluated with the MFMA dense kernel (tanh fused)."""

    def forward(self, x):
        seg = x if isinstance(x, _hip.SegInput) else None      # input rows described by their segments (read in place by the int8-slice kernel)
        if seg is not None:
            mods = list(self)
            i8 = (len(mods) == 3 and isinstance(mods[1], nn.Tanh) and seg.dtype == torch.float64 and MLP_MATRIX_ARITHMETIC_F64[0] != "f64"
                  and mods[0].weight.dtype == torch.float64 and MLP_I8_MIN_COLS[0] <= mods[2].out_features and seg.shape[0] >= MLP_I8_MIN_ROWS[0]
                  and mods[0].in_features <= _hip.MLP2_I8_MAX_IN and mods[0].out_features <= _hip.MLP2_MAX_HIDDEN and mods[0].out_features % 4 == 0
                  and not torch.is_grad_enabled())
            if not i8:
                x = seg.materialize()
                seg = None
        _hip.require_device(*(seg.tensors() if seg is not None else [x]))
        mods = list(self)
        if seg is None and autograd._needs_grad(x, *self.parameters()):
            if (len(mods) == 3 and isinstance(mods[1], nn.Tanh) and autograd.mlp2_small_ok(x, mods[0], mods[2])
                    and mods[0].weight.dtype == x.dtype):
                # narrow head (e.g. 4 -> 128 -> 10 of an 'f' layer) on data rows: fused forward, one-launch backward
                return autograd.Mlp2SmallFn.apply(x, mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias)
            # training: one dense launch per layer wrapped in autograd (the hidden activations are what backward needs)
            i = 0
            while i < len(mods):
                lin = mods[i]
                act = 1 if (i + 1 < len(mods) and isinstance(mods[i + 1], nn.Tanh)) else 0
                w, b = lin.weight, lin.bias
                if w.dtype != x.dtype:
                    w, b = w.to(x.dtype), b.to(x.dtype)
                x = autograd.linear(x, w, b, act)
                i += 2 if act else 1
            return x
        if (len(mods) == 3 and isinstance(mods[1], nn.Tanh) and mods[0].in_features <= _hip.MLP2_MAX_IN
                and mods[0].out_features <= _hip.MLP2_MAX_HIDDEN and mods[0].out_features % 4 == 0):
            # Linear-tanh-Linear (the reference's default "128"): one fused launch, the hidden activations never leave the registers
            ps = [mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias]
            if ps[0].dtype != x.dtype:
                ps = [p.to(x.dtype) for p in ps]
            ps = [p.detach() for p in ps]
            if (x.dtype == torch.float64 and MLP_MATRIX_ARITHMETIC_F64[0] != "f64" and MLP_I8_MIN_COLS[0] <= ps[2].shape[0] <= (1 << 20)
                    and x.shape[0] >= MLP_I8_MIN_ROWS[0] and mods[0].in_features <= _hip.MLP2_I8_MAX_IN):
                # wide float64 output (the 548-column parameter block of an e4 block): the float64 matrix cores run at the float64 vector rate, so
                # the second product goes to the int8 matrix cores as an error-free product of digit slices (csrc/mlp_i8_kernels.hip: 2.9 -> 1.2 ms
                # per 2^20 rows of the C3 block); the digit image of the weights is rebuilt when they change
                slices = 6 if MLP_MATRIX_ARITHMETIC_F64[0] == "i8x6" else 5
                lin = mods[2]
                key = (lin.weight._version, lin.weight.data_ptr(), lin.bias._version, lin.bias.data_ptr(), str(lin.weight.device), slices)
                hit = getattr(self, "_i8_image", None)
                if hit is None or hit[0] != key:
                    # non-finite weights cannot be cut into digits: they take the exact path, which propagates them
                    ok = bool(torch.isfinite(ps[2]).all()) and bool(torch.isfinite(ps[3]).all())
                    hit = (key, _hip.mlp2_i8_pack(ps[2], ps[3], slices) if ok else None)
                    self._i8_image = hit
                if hit[1] is not None:
                    return _hip.mlp2_i8(x, ps[0], ps[1], hit[1], ps[2].shape[0], slices)
            x = _hip.as_matrix(x)                                # (non-finite weights: the exact kernel)
            if x.dtype == torch.float32 and ps[2].shape[0] >= 256 and x.shape[0] >= 4096:
                # wide float32 output (the 548-column parameter block of an e4 block: sampling, the two-launch log-prob path): the second layer on
                # split-bf16 MFMA (jf_linear_split_f32, 0.23 ms per 2^18 rows) after a streaming first layer beats the fused exact-f32 MFMA launch
                # (jf_mlp2: 0.36 .. 0.39 ms per 2^18 rows) although the hidden activations make a round trip through HBM
                h = _hip.linear(x, ps[0], ps[1], 1)
                if _hip.linear_split_ok(h, ps[2], ps[3]):
                    return _hip.linear_split(h, ps[2], ps[3])
                return _hip.linear(h, ps[2], ps[3], 0)
            return _hip.mlp2(x, *ps)
        i = 0
        while i < len(mods):
            lin = mods[i]
            assert isinstance(lin, nn.Linear)
            act = 1 if (i + 1 < len(mods) and isinstance(mods[i + 1], nn.Tanh)) else 0
            w, b = lin.weight, lin.bias
            if w.dtype != x.dtype:
                w, b = w.to(x.dtype), b.to(x.dtype)
            x = _hip.linear(x, w.detach(), b.detach(), act)
            i += 2 if act else 1
        return x


def _manifold_family(layers):
    """kernel family letter if the layers of a block can run as ONE manifold-chain launch, else None."""
    from ..layers.intervals.rational_quadratic_spline import rational_quadratic_spline
    from ..layers.spheres.sphere_base import sphere_base
    if not (1 <= len(layers) <= _hip.JF_MAX_MCHAIN):
        return None
    if all(type(l) is rational_quadratic_spline for l in layers):
        return "r"
    l0 = layers[0]
    if isinstance(l0, sphere_base) and l0.FAMILY is not None and all(type(l) is type(l0) for l in layers):
        if any(l.always_parametrize_in_embedding_space for l in layers):
            return None
        return l0.FAMILY
    return None


def _layer_groups(layers):
    """split a block's layers into launch groups: maximal runs of chainable 'g' layers (one fused launch each), every other layer alone"""
    groups, run = [], []
    for l in layers:
        if (type(l) is gfl.gf_block and l.dimension <= _hip.GF_MAX_DIM and len(run) < _hip.JF_MAX_CHAIN
                and gfl.chain_fits(run + [l])):          # wide layers: a run ends where one launch would no longer fit a CU's LDS
            run.append(l)
            continue
        if run:
            groups.append(run)
            run = []
        if type(l) is gfl.gf_block:
            run = [l]
        else:
            groups.append([l])
    if run:
        groups.append(run)
    return groups


def _manifold_chain(fam, layers, direction, x, log_det, extra, only_last_first, x_out, base_logp_in, want_base_logp, status, pre_ld=None, pre_blp=None):
    structs = []
    for l in layers:
        if fam == "r":
            structs.append(l.c_struct())
        else:
            structs.append(l.c_struct(1 if (only_last_first or l.euclidean_to_sphere_as_first) else 0))
    if extra is None:
        rows = [l._params_for(x, None) for l in layers]
        params = torch.cat(rows, dim=1) if len(rows) > 1 else rows[0]
    else:
        params = extra
    return _hip.mchain(fam, direction, x, log_det, params, structs, layers[0].dimension, x_out=x_out, base_logp_in=base_logp_in,
                       want_base_logp=want_base_logp, status=status, pre_ld=pre_ld, pre_blp=pre_blp)


class pdf(nn.Module):
    def __init__(self,
                 pdf_defs,
                 flow_defs,
                 options_overwrite=dict(),
                 conditional_input_dim=None,
                 amortization_mlp_dims="128",
                 predict_log_normalization=False,
                 join_poisson_and_pdf_description=False,
                 hidden_mlp_dims_poisson="128",
                 rank_of_mlp_mappings_poisson=0,
                 amortization_mlp_use_custom_mode=False,
                 amortization_mlp_ranks=0,
                 amortization_mlp_highway_mode=0,
                 amortize_everything=False,
                 use_as_passthrough_instead_of_pdf=False,
                 skip_mlp_initialization=False,
                 verbose=False):
        """Arguments as documented for the reference (:63-100)."""
        super().__init__()
        self.amortization_mlp_use_custom_mode = amortization_mlp_use_custom_mode
        # Poisson head (:51, 104-110, 467-477, 675-716, 836-877): the pdf also predicts the log-mean of a Poisson count -- a parameter when the
        # pdf is unconditional, the LAST output column of the first amortisation MLP with join_poisson_and_pdf_description, or (the reference's
        # "outdated" separate form: built for state_dict compatibility, log_mean_poisson raises for it as the reference's does) an MLP of its own
        self.predict_log_normalization = bool(predict_log_normalization)
        if self.predict_log_normalization:
            assert not amortize_everything, ("Log Poisson prediction works only without full amortization in the default PDF. It can be used in the "
                                             "*fully_amortized_pdf*!")
        self.hidden_mlp_dims_poisson = hidden_mlp_dims_poisson
        self.rank_of_mlp_mappings_poisson = rank_of_mlp_mappings_poisson
        self.join_poisson_and_pdf_description = join_poisson_and_pdf_description
        self.amortization_mlp_highway_mode = amortization_mlp_highway_mode
        self.amortize_everything = amortize_everything
        self.use_as_passthrough_instead_of_pdf = use_as_passthrough_instead_of_pdf
        self.skip_mlp_initialization = skip_mlp_initialization
        self.total_number_amortizable_params = None
        if amortize_everything:
            assert amortization_mlp_use_custom_mode, "Amortizing all MLPs requires custom MLPs."
            self.total_number_amortizable_params = 0
        # kernel status words (rows with out-of-range spline inputs, non-finite results, non-converged Newton rows) -> the reference's
        # exceptions / warnings.  True (default) / "immediate": checked before the call returns, so the exception belongs to the batch that
        # caused it, like the reference's (spline_fns.py:57-59, default.py:1516).  "deferred": log-prob calls copy the words to pinned host memory
        # asynchronously and raise at the next call or in flush_status() (saves one host-device round trip per call; throughput loops
        # such as bench.py opt into it and flush inside their timed region).  False: never.
        self.check_status = True
        self._capture_status = None
        self.fold_combine = os.environ.get("JF_FOLD_COMBINE", "1") != "0"    # the last fused block adds the per-block sums itself (no jf_combine_rows launch)
        self.force_fused_manifold_blocks = False     # tests: run jf_cond_<fam>_chain_inv also where the two-launch path is the faster default
        # conditional e-blocks (Linear-tanh-Linear MLP + g layers, D in {3,4}, float32) as ONE launch with the parameter block kept on chip
        # (jf_cond_gf_chain_inv): +9 % on the C3 step against jf_mlp2 + jf_gf_chain_inv.  False selects the two-launch path.
        self.fuse_conditional_blocks = True
        # matrix arithmetic of the fused block's 128 -> P product: "split_f16" (two f16 pieces per f32 operand, operands scaled into the normal
        # f16 range, three MFMA passes with f32 accumulation: representation error <= 2^-22 per operand, below the rounding of the f32
        # accumulation itself), "split_bf16" (three bf16 pieces, six passes: products exact, result within ~3 * 2^-24 relative of the f32 dot
        # product) -- both keep the parameters in registers -- or "f32" (exact f32-input MFMA, parameter tile in LDS).  Layer options outside
        # the split kernel's set fall back to "f32" by themselves.
        # JF_FUSED_MATRIX_ARITHMETIC=f32 in the environment selects the exact-f32 kernel process-wide (an operational fallback while the
        # full-batch hazard of DESIGN.md 3.9 has a remedy but no root cause).
        self.fused_matrix_arithmetic = os.environ.get("JF_FUSED_MATRIX_ARITHMETIC", "split_f16")
        # which kernel runs the split-bf16 fused block: "auto" = "split" (cond_split_kernels.hip).  (The persistent ping-pong variant of round 3
        # measured the same and left the product in round 5: scripts/probe/cond_pp/.)
        self.fused_block_kernel = os.environ.get("JF_FUSED_BLOCK_KERNEL", "auto")
        if self.fused_block_kernel not in ("auto", "split"):
            raise ValueError("JF_FUSED_BLOCK_KERNEL must be 'auto' or 'split', got %r" % self.fused_block_kernel)
        if self.fused_matrix_arithmetic not in ("split_f16", "split_bf16", "f32"):
            raise ValueError("JF_FUSED_MATRIX_ARITHMETIC must be 'split_f16', 'split_bf16' or 'f32', got %r" % self.fused_matrix_arithmetic)
        self._packed_cache = {}
        self._step_plans = {}
        # side streams a recorded step may use for its (independent) blocks at batches up to plan_lane_max_rows.  Default 1 = none: measured on C3,
        # 2^15 .. 2^18 rows, three lanes stretch every kernel (each fills the chip on its own) and the step gets no shorter
        self.plan_lanes = int(os.environ.get("JF_PLAN_LANES", "1"))
        self.plan_lane_max_rows = int(os.environ.get("JF_PLAN_LANE_MAX_ROWS", str(1 << 18)))
        # the same idea without streams: the blocks after the first are launched without the queue's barrier bit (csrc/plan.hip any_order)
        self.plan_overlap_blocks = os.environ.get("JF_PLAN_OVERLAP", "0") == "1"
        self.plan_overlap_max_rows = int(os.environ.get("JF_PLAN_OVERLAP_MAX_ROWS", str(1 << 40)))
        # gradient mode, float64: chains on a low-rank last MLP stage never materialise their (B, P) parameter / gradient blocks
        self.lowrank_chain_training = os.environ.get("JF_LOWRANK_CHAIN_TRAINING", "1") != "0"
        self.use_step_plans = os.environ.get("JF_STEP_PLANS", "0") == "1"      # forward() through recorded step plans (planned_forward)
        # float32 log-prob steps of up to merge_max_rows rows that start with an unconditional broadcast g chain and a one-launch `f` block issue
        # the two as ONE launch (csrc/merged_kernels.hip: they overlap each other's latency -- 0.055 -> 0.037 ms at 2^17 rows, 0.228 -> 0.210 at
        # 2^20); 0 = never
        self.merge_max_rows = int(os.environ.get("JF_MERGE_MAX_ROWS", str(1 << 40)))
        self.merge_max_rows_pipelined = int(os.environ.get("JF_MERGE_MAX_ROWS_PIPELINED", str(1 << 19)))     # (see PipelinedForward)
        self._merge_ok = {}
        # gradient mode: the blocks of a training step are independent given the targets too.  With train_streams > 1 every block's forward is
        # issued on one of that many streams (torch.autograd runs a node's backward on the stream of its forward), so the latency-bound kernels
        # of one block's adjoint overlap the other blocks'; the per-block log-dets / base log-probs are added at the end instead of threaded.
        # Two streams by default (round 5, after the adjoint kernels of the side blocks got shorter): C3 training 1.360 -> 1.303 ms per step,
        # C5 1.665 -> 1.655; three streams give nothing more (1.367 / 1.650).  1 = everything on the caller's stream.
        self.train_streams = int(os.environ.get("JF_TRAIN_STREAMS", "2"))
        self._train_stream_objs = {}

        self._read_model_definition(pdf_defs, flow_defs, options_overwrite, conditional_input_dim, amortization_mlp_dims,
                                    amortization_mlp_ranks)
        self._init_flow_structure()
        self._init_encoding_structure()
        self.init_params()

    # =========================================================================================== construction
    def _read_model_definition(self, pdf_defs, flow_defs, options_overwrite, conditional_input_dim, mlp_dims, mlp_ranks):
        """option resolution: (sub,layer) tuple > sub-pdf index > layer letter (:153-325)."""
        self.pdf_defs_list = pdf_defs.split("+")
        self.flow_defs_list = ["".join(canonical(c) for c in f) for f in flow_defs.split("+")]
        if len(self.pdf_defs_list) != len(self.flow_defs_list):
            raise Exception("PDF defs list has to be same length as flow defs list, but ... ", self.pdf_defs_list, self.flow_defs_list)
        nsub = len(self.pdf_defs_list)
        self.flow_opts = dict()
        for si, letters in enumerate(self.flow_defs_list):
            self.flow_opts[si] = []
            for li, letter in enumerate(letters):
                opts = obtain_default_options(letter)
                chosen = None
                for k, v in options_overwrite.items():
                    if type(k) == tuple:
                        assert type(k[0]) == int and type(k[1]) == int, "Require 2 ints for tuple-based flow definition!"
                        assert 0 <= k[0] < nsub, "Index of detailed options is outside allowed range of defined autoregressive structure."
                        if k == (si, li):
                            assert len(v) == 1 and canonical(list(v.keys())[0]) == letter
                            chosen = list(v.values())[0]
                if chosen is None:
                    for k, v in options_overwrite.items():
                        if type(k) == int:
                            assert 0 <= k < nsub, "Index of detailed options is outside allowed range of defined autoregressive structure."
                            if k == si:
                                for kk, vv in v.items():
                                    if canonical(kk) == letter:
                                        chosen = vv
                if chosen is None:
                    for k, v in options_overwrite.items():
                        if type(k) == str and canonical(k) == letter:
                            chosen = v
                if chosen is not None:
                    for name, val in chosen.items():
                        check_flow_option(letter, name, val)
                        opts[name] = val
                self.flow_opts[si].append(opts)

        self.conditional_input_dim = conditional_input_dim
        self.encoding_type = "single"
        if type(conditional_input_dim) == list:
            assert all(type(ci) == int for ci in conditional_input_dim)
            self.encoding_type = "multi"
        if type(mlp_dims) == str:
            mlp_dims = [mlp_dims] * nsub
        elif type(mlp_dims) != list:
            raise Exception("Hidden MLP dimensions must be defined either str or list, received ", type(mlp_dims))
        if len(mlp_dims) != nsub:
            raise Exception("hidden mlp dimension definitions for sub pdfs is wrong length (%d) .. requires length (%d)" % (len(mlp_dims), nsub))
        self.amortization_mlp_dims = mlp_dims
        if type(mlp_ranks) in (int, str):
            mlp_ranks = [mlp_ranks] * nsub
        elif type(mlp_ranks) != list:
            raise Exception("Rank of MLP sub pdfs has to defined as an int or list type!")
        self.amortization_mlp_ranks = mlp_ranks
        self.force_permanent_parameters_in_first_subpdf = 1 if (conditional_input_dim is None and not self.amortize_everything) else 0

    def _init_flow_structure(self):
        """instantiate the layers (:378-479)."""
        self.layer_list = nn.ModuleList()
        self.num_parameter_list = []
        for si, sub in enumerate(self.pdf_defs_list):
            kind = sub[0]
            dim = int(sub.split("_")[0][1:])
            letters = self.flow_defs_list[si]
            block = nn.ModuleList()
            for li, letter in enumerate(letters):
                if opts_dict[letter]["type"] != kind:
                    raise Exception("layer type ", letter, " is not compatible with flow type ", sub)
                kw = copy.deepcopy(self.flow_opts[si][li])
                kw["use_permanent_parameters"] = 1 if (self.force_permanent_parameters_in_first_subpdf and si == 0) else 0
                first = 1 if (li == 0 and not self.use_as_passthrough_instead_of_pdf) else 0
                if kind == "s":
                    kw["euclidean_to_sphere_as_first"] = first
                elif kind == "i":
                    bounds = sub.split("_")[1:]
                    kw["low_boundary"] = float(bounds[0]) if bounds else 0.0
                    kw["high_boundary"] = float(bounds[1]) if bounds else 1.0
                    kw["euclidean_to_interval_as_first"] = first
                elif kind == "e" and letter != "x":
                    if li == len(letters) - 1 and kw["skip_model_offset"] == 0:
                        kw["model_offset"] = 1
                    elif li == 0 and letter == "g":
                        if kw["replace_first_sigmoid_with_icdf"] > 0 and kw["inverse_function_type"] == "isigmoid":
                            kw["inverse_function_type"] = "inormal_partly_precise"
                kw.pop("skip_model_offset", None)
                kw.pop("replace_first_sigmoid_with_icdf", None)
                block.append(layer_class(letter)(dim, **kw))
            self.layer_list.append(block)
            self.num_parameter_list.append([l.get_total_param_num() for l in block])
        self.log_normalization = None
        if self.predict_log_normalization:
            assert len(self.pdf_defs_list) == 1, ("You chose to predict log-lambda, which is only allowed with a single sub-pdf (no autoregressive "
                                                  "structure). For autoregressive PDFs with log-lambda prediction, use fully amortized PDFs.")
            if self.force_permanent_parameters_in_first_subpdf:
                self.log_normalization = nn.Parameter(torch.randn(1).unsqueeze(0))
            else:
                self.log_normalization = torch.zeros(1).unsqueeze(0)
        self.update_embedding_structure()

    def get_embedding_flags(self):
        flags = []
        for block in self.layer_list:
            f = block[0].always_parametrize_in_embedding_space
            assert all(l.always_parametrize_in_embedding_space == f for l in block)
            flags.append(f)
        return flags

    def set_embedding_flags(self, usement_flag, sub_pdf_index=None):
        """switch (sub-)manifolds between intrinsic and embedding default coordinates (:346-374)."""
        assert usement_flag in (True, False)
        for si, block in enumerate(self.layer_list):
            if sub_pdf_index is None or si == sub_pdf_index:
                for l in block:
                    l.always_parametrize_in_embedding_space = usement_flag
        self.update_embedding_structure()

    def update_embedding_structure(self):
        """column bookkeeping of target / base tensors (:481-567)."""
        self.target_dims_intrinsic, self.target_dims_embedded, self.target_dims = [], [], []
        self.target_dim_indices_intrinsic, self.target_dim_indices_embedded, self.target_dim_indices, self.base_dim_indices = [], [], [], []
        ti = te = td = tb = 0
        for block in self.layer_list:
            intr = block[-1].get_layer_intrinsic_target_dimension()
            emb = block[-1].get_layer_embedded_target_dimension()
            use_emb = any(l.always_parametrize_in_embedding_space for l in block)
            base = block[0].get_layer_base_dimension()
            cur = emb if use_emb else intr
            self.target_dims_intrinsic.append(intr)
            self.target_dims_embedded.append(emb)
            self.target_dims.append(cur)
            self.base_dim_indices.append((tb, tb + base)); tb += base
            self.target_dim_indices_intrinsic.append((ti, ti + intr)); ti += intr
            self.target_dim_indices_embedded.append((te, te + emb)); te += emb
            self.target_dim_indices.append((td, td + cur)); td += cur
        self.total_target_dim_intrinsic, self.total_target_dim_embedded, self.total_target_dim, self.total_base_dim = ti, te, td, tb

    def _init_encoding_structure(self):
        """one amortisation MLP per sub-pdf that has parameters and a non-empty input (:571-722)."""
        self.mlp_predictors = nn.ModuleList()
        self.log_normalization_mlp = None
        if self.skip_mlp_initialization:
            if self.predict_log_normalization:          # external MLPs must predict the Poisson and pdf description jointly (:717-721)
                assert self.join_poisson_and_pdf_description
            return
        prev = 0
        for si in range(len(self.pdf_defs_list)):
            emb = self.layer_list[si][-1]._embedding_conditional_return_num()
            npar = sum(self.num_parameter_list[si])
            if self.predict_log_normalization and si == 0 and self.join_poisson_and_pdf_description and self.conditional_input_dim is not None:
                npar += 1                                   # log-lambda = the last output of the first MLP (:624-627)
            if si == 0 and self.conditional_input_dim is None:
                self.mlp_predictors.append(None)
                if self.amortize_everything:
                    self.total_number_amortizable_params += npar
            elif npar == 0:
                self.mlp_predictors.append(None)
            else:
                in_dim = prev
                if self.conditional_input_dim is not None:
                    in_dim += self.conditional_input_dim if type(self.conditional_input_dim) == int else self.conditional_input_dim[si]
                if self.amortization_mlp_use_custom_mode:
                    mlp = AmortizableMLP(in_dim, list_from_str(self.amortization_mlp_dims[si]), npar,
                                         low_rank_approximations=self.amortization_mlp_ranks[si],
                                         use_permanent_parameters=not self.amortize_everything,
                                         highway_mode=self.amortization_mlp_highway_mode, svd_mode="smart")
                    if self.amortize_everything:
                        self.total_number_amortizable_params += mlp.num_amortization_params
                else:
                    hidden = list_from_str(self.amortization_mlp_dims[si])
                    dims_in = [in_dim] + hidden
                    dims_out = hidden + [npar]
                    mods = []
                    for i in range(len(dims_in)):
                        mods.append(nn.Linear(dims_in[i], dims_out[i]))
                        if i < len(dims_in) - 1:
                            mods.append(nn.Tanh())
                    mlp = HipLinearStack(*mods)
                self.mlp_predictors.append(mlp)
            prev += emb
        if self.predict_log_normalization and self.conditional_input_dim is not None and not self.join_poisson_and_pdf_description:
            # a predictor of its own (:675-716) -- same modules as the reference builds, so that its state_dict loads
            in_dim = self.conditional_input_dim if type(self.conditional_input_dim) == int else self.conditional_input_dim[0]
            if self.amortization_mlp_use_custom_mode:
                self.log_normalization_mlp = AmortizableMLP(in_dim, self.hidden_mlp_dims_poisson, 1, low_rank_approximations=self.rank_of_mlp_mappings_poisson,
                                                            use_permanent_parameters=True, highway_mode=self.amortization_mlp_highway_mode, svd_mode="smart")
            else:
                hidden = list_from_str(self.amortization_mlp_dims[0])
                dims_in, dims_out = [in_dim] + hidden, hidden + [1]
                mods = []
                for i in range(len(dims_in)):
                    mods.append(nn.Linear(dims_in[i], dims_out[i]))
                    if i < len(dims_in) - 1:
                        mods.append(nn.Tanh())
                    else:
                        mods[-1].weight.data /= 1000.0
                        mods[-1].bias.data[0] = -1.0
                self.log_normalization_mlp = HipLinearStack(*mods)

    def log_mean_poisson(self, conditional_input=None, amortization_parameters=None):
        """log-lambda of the Poisson head, (B, 1) (or the (1, 1) parameter of an unconditional pdf)  (:836-877)."""
        if self.log_normalization is None:
            raise Exception("This PDF does not predict the log-mean of a Poisson distriution. Initialize with 'predict_log_normalization'=True for this "
                            "possibility.")
        if amortization_parameters is not None:
            assert amortization_parameters.shape[1] == self.total_number_amortizable_params
        if conditional_input is None:
            if amortization_parameters is not None:
                raise Exception("Currently there is no support for the prediction of log-lambda and simultanesouly passing amortization_parameters")
            return self.log_normalization
        if self.join_poisson_and_pdf_description:
            mlp = self.mlp_predictors[0]
            if amortization_parameters is not None:
                assert isinstance(mlp, AmortizableMLP) and not mlp.use_permanent_parameters
                return mlp(conditional_input, extra_inputs=amortization_parameters[:, :mlp.num_amortization_params])[:, -1:]
            return mlp(conditional_input)[:, -1:]           # the last parameter of the first MLP is log-lambda (convention)
        raise NotImplementedError("This way of independenly predicting the normalization (from all other parameters) is outdated!")

    def init_params(self, data=None, damping_factor=1000.0, mvn_min_max_sv_ratio=1e-4):
        """layer "desired init" vectors become the final bias of each MLP (everything else / damping_factor) or are written into
        the permanent parameters (:1817-1952).  With `data` (B, total_target_dim) the Euclidean blocks are initialised from the data
        (means, principal axes, percentiles; init_fns.find_init_pars_of_chained_blocks), the data passing through each initialised layer's
        HIP kernel on the way."""
        if data is not None:
            assert data.shape[1] == self.total_target_dim, "Initialization with data must match the target dimension of the PDF!"
        global_init = torch.zeros(self.total_number_amortizable_params) if self.amortize_everything else None
        gi = 0
        dim_index = 0
        module_device = self.obtain_current_dtype_n_device()[1]
        with torch.no_grad():
            for si, block in enumerate(self.layer_list):
                this_dim = self.target_dims[si]
                if self.pdf_defs_list[si][0] == "e":
                    these = init_fns.find_init_pars_of_chained_blocks(list(block), data[:, dim_index:dim_index + this_dim] if data is not None else None,
                                                                      mvn_min_max_sv_ratio=mvn_min_max_sv_ratio)
                else:
                    parts = [l.get_desired_init_parameters() for l in block]
                    these = torch.cat(parts) if len(parts) else torch.zeros(0)
                dim_index += this_dim
                if module_device is not None:
                    these = these.to(module_device)          # the init vector lands in parameters: keep them on the module's device
                if len(these) == 0:
                    continue
                mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
                if mlp is not None and self.predict_log_normalization and self.join_poisson_and_pdf_description and si == 0:
                    these = torch.cat([these, torch.tensor([0.1], dtype=these.dtype, device=these.device)])      # log-lambda at initialisation (:1893-1896)
                if mlp is not None:
                    if isinstance(mlp, AmortizableMLP):
                        if self.amortize_everything:
                            n = mlp.num_amortization_params
                            global_init[gi:gi + n] = mlp.obtain_default_init_tensor(fix_final_bias=these.cpu(), prev_damping_factor=damping_factor)
                            gi += n
                        else:
                            mlp.initialize_uvbs(fix_final_bias=these.cpu(), prev_damping_factor=damping_factor)
                    else:
                        for m in mlp:
                            if hasattr(m, "weight"):
                                nn.init.kaiming_uniform_(m.weight.data, a=numpy.sqrt(5))
                                fan_in, _ = nn.init._calculate_fan_in_and_fan_out(m.weight.data)
                                bound = 1 / numpy.sqrt(fan_in)
                                nn.init.uniform_(m.bias.data, -bound, bound)
                                m.weight.data /= damping_factor
                                m.bias.data /= damping_factor
                        mlp[-1].bias.data = these.data.type(mlp[-1].bias.data.dtype)
                else:
                    c = 0
                    for l in block:
                        n = l.get_total_param_num()
                        if not self.amortize_everything:
                            l.init_params(these[c:c + n])
                        c += n
                    if self.amortize_everything:
                        global_init[gi:gi + c] = these.cpu()
                        gi += c
        return global_init

    def count_parameters(self, verbose=False):
        """number of trainable parameters (MLPs + permanent layer parameters) (:724-830)."""
        tot = 0
        for m in self.mlp_predictors:
            if m is not None:
                tot += sum(int(numpy.prod(p.size())) for p in m.parameters() if p.requires_grad)
        if self.log_normalization_mlp is not None:
            tot += sum(int(numpy.prod(p.size())) for p in self.log_normalization_mlp.parameters() if p.requires_grad)
        for block in self.layer_list:
            for l in block:
                tot += sum(int(numpy.prod(p.size())) for p in l.parameters() if p.requires_grad)
        if verbose:
            print("total Conditional PDF pars: %d" % tot)
        return tot

    def get_total_embedding_dim(self):
        return sum(block[-1]._embedding_conditional_return_num() for block in self.layer_list)

    def obtain_current_dtype_n_device(self):
        try:
            first = next(self.parameters())
        except StopIteration:
            return None, None
        return first.dtype, first.device

    # =========================================================================================== parameter routing
    def _conditioning_rows(self, x, data_summary):
        """log-prob direction: every target is known up front, so the input row of every amortisation MLP,
        cat[conditional_input, embed(x_0), embed(x_1), ...] (:946-962), is a list of SEGMENTS of the caller's tensors: column ranges copied as they
        are, S1 / S2 angles embedded (sphere_base.py:786-794).  Block si reads the first ``prefix[si]`` columns.  Nothing is launched here: a block
        whose prefix is one plain column range reads that view, a consumer kernel with segment support reads the segments in place
        (_hip.SegInput), anything else materialises the rows once (jf_conditioning_rows).  None when nothing needs them / a per-block summary list."""
        if type(data_summary) == list or len(self.mlp_predictors) == 0 or all(m is None for m in self.mlp_predictors):
            return None
        segs, prefix, width = [], [], 0
        if data_summary is not None:
            segs.append((data_summary, 0))
            width = data_summary.shape[1]
        for si, block in enumerate(self.layer_list):
            prefix.append(width)
            a, b = self.target_dim_indices[si]
            last = block[-1]
            kind = 0
            if self.pdf_defs_list[si][0] == "s" and (b - a) == last.dimension:
                kind = last.dimension                                            # intrinsic angles -> embedding (sphere_base.py:786-794)
            segs.append((x[:, a:b], kind))
            width += last._embedding_conditional_return_num()
        if len(segs) > _hip.JF_MAX_SEGMENTS:
            return None
        return {"segs": segs, "prefix": prefix, "cond": data_summary, "B": x.shape[0], "dtype": x.dtype, "device": x.device, "inputs": {}}

    def _mlp_input(self, si, data_summary, embeds):
        if isinstance(embeds, dict):
            n = embeds["prefix"][si]
            if n == 0:
                raise Exception("extra conditional input is empty but required for encoding!")
            hit = embeds["inputs"].get(n)
            if hit is None:
                keep, w = [], 0
                for t, kind in embeds["segs"]:                                   # the segments that make up the first n columns
                    if w >= n:
                        break
                    keep.append((t, kind))
                    w += t.shape[1] if kind == 0 else kind + 1
                assert w == n, (w, n)
                # neighbouring plain column ranges of one tensor are one range (x[:, 0:4] + x[:, 4:8] = x[:, 0:8])
                merged = []
                for t, kind in keep:
                    if (merged and kind == 0 and merged[-1][1] == 0 and t.dim() == 2 and merged[-1][0].stride() == t.stride()
                            and merged[-1][0].data_ptr() + merged[-1][0].shape[1] * t.element_size() * t.stride(1) == t.data_ptr()):
                        p = merged[-1][0]
                        merged[-1] = (torch.as_strided(p, (p.shape[0], p.shape[1] + t.shape[1]), p.stride(), p.storage_offset()), 0)
                    else:
                        merged.append((t, kind))
                if len(merged) == 1 and merged[0][1] == 0:
                    hit = merged[0][0]                                           # one plain column range: the view itself
                else:
                    hit = _hip.SegInput(merged, embeds["B"], embeds["dtype"], embeds["device"])
                embeds["inputs"][n] = hit
            return hit
        if data_summary is not None:
            inp = data_summary[si] if type(data_summary) == list else data_summary
            if len(embeds) > 0:
                inp = torch.cat([inp] + embeds, dim=1)
            return inp
        if len(embeds) > 0:
            return torch.cat(embeds, dim=1) if len(embeds) > 1 else embeds[0]
        raise Exception("extra conditional input is empty but required for encoding!")

    def _fusable_block(self, si, layers, only_last, amort, dtype):
        """can sub-pdf si run as ONE fused launch (amortisation MLP + its g layers, parameter block kept on chip)?"""
        if self._poisson_column(si):                       # the MLP emits one more column than the block has parameters
            return None
        if not self.fuse_conditional_blocks or only_last or amort is not None or _hip.BINS_LOG is not None:
            return None
        if dtype != torch.float32:      # measured: in float64 the two-launch path (jf_mlp2 + jf_gf_chain_inv) is faster
            return None
        mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
        if not isinstance(mlp, HipLinearStack) or len(mlp) != 3 or not isinstance(mlp[1], nn.Tanh):
            return None
        if not (3 <= layers[0].dimension <= 4) or not gfl.chain_supported(layers):
            return None
        if any(l.nonlinear_stretch_type != "classic" or l.has_extended_options for l in layers):
            return None
        if mlp[0].in_features > _hip.COND_GF_MAX_IN or mlp[0].out_features > _hip.COND_GF_MAX_HIDDEN or mlp[0].out_features % 4:
            return None
        ps = [mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias]
        if ps[0].dtype != dtype:
            ps = [p.to(dtype) for p in ps]
        return [p.detach() for p in ps]

    def _fusable_manifold_block(self, si, layers, only_last, amort, dtype):
        """sub-pdf si = default amortisation MLP (Linear-tanh-Linear) + one chain of 'r' / 'o' / 'm' / 'f' layers with <= 64 parameters per
        row: (family, [w1, b1, w2, b2]) for jf_cond_<fam>_chain_inv, else None"""
        if self._poisson_column(si):                       # the MLP emits one more column than the block has parameters
            return None
        if not self.fuse_conditional_blocks or only_last or amort is not None or _hip.BINS_LOG is not None:
            return None
        mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
        if not isinstance(mlp, HipLinearStack) or len(mlp) != 3 or not isinstance(mlp[1], nn.Tanh):
            return None
        fam = _manifold_family(layers)
        if fam is None or fam not in _hip.COND_MCHAIN_FAMILIES:
            return None
        # measured on 2^20 rows (scripts/bench_configs.py): the one-launch form wins for float32 'f' blocks (0.146 ms vs 0.16 + 0.05 ms); for the
        # 'r' / 'o' / 'm' families and for float64 the two launches (resident narrow-output jf_mlp2 + chain) are faster (f64 'o': 0.48 vs 0.96 ms)
        # (re-measured at the end of round 4, C4's float32 'o' block: one launch 0.171 ms, jf_mlp2 0.117 + jf_o_chain_inv 0.041 -- scripts/probe/c4_fused_ab.py)
        plain_f = fam == "f" and not any(l._vertical or l._circular for l in layers)      # with nested spline flows: 0.57 vs 0.21 + 0.28 ms
        if not (plain_f and dtype == torch.float32) and not self.force_fused_manifold_blocks:
            return None
        if mlp[0].in_features > _hip.COND_GF_MAX_IN or mlp[0].out_features > _hip.COND_GF_MAX_HIDDEN or mlp[2].out_features > _hip.COND_MCHAIN_MAX_PARAMS:
            return None
        if fam == "f" and any(getattr(l, "add_correlated_rq_spline_flow", 0) for l in layers):
            return None
        ps = [mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias]
        if ps[0].dtype != dtype:
            ps = [p.to(dtype) for p in ps]
        return fam, [p.detach() for p in ps]

    def _fusable_lowrank_block(self, si, layers, only_last, amort, like):
        """sub-pdf si = a two-stage AmortizableMLP with a low-rank last stage + chainable g layers at default options: the weight views for
        jf_amlp_gf_chain_inv (v1, u1, b1, v2, u2, b2), else None"""
        if self._poisson_column(si):                       # the MLP emits one more column than the block has parameters
            return None
        if not self.fuse_conditional_blocks or only_last or amort is not None or _hip.BINS_LOG is not None:
            return None
        mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
        if not isinstance(mlp, AmortizableMLP) or mlp.highway_mode != 0 or not mlp.use_permanent_parameters or len(mlp.stages or []) != 2:
            return None
        s1, s2 = mlp.stages
        if s2["full"] or s2["rank"] > 16 or (not s1["full"] and s1["rank"] > 16) or s1["inp"] > 32 or s1["out"] > 128:
            return None
        if not gfl.chain_supported(layers) or layers[0].dimension > 8:
            return None
        for l in layers:
            c = l.c_struct()
            if l.has_extended_options:
                return None
            if not (c.num_kde == 10 and c.hh_iter <= 8 and c.nonlinear_stretch_type == _hip.GF_STRETCH_CLASSIC and c.width_mode == _hip.GF_WIDTH_SMOOTH
                    and not c.clamp_widths and c.fit_normalization and c.regulate_normalization):
                return None
        return mlp.lowrank_views(mlp._flat(like))

    def _lowrank_chain_ok(self, si, layers, only_last, amort, x, mlp):
        """gradient mode: sub-pdf si can run autograd.LowRankGfChainFn (float64, last MLP stage low-rank with rank <= 8, <= 8 dimensions,
        g layers at default options; pdf.lowrank_chain_training = False or JF_LOWRANK_CHAIN_TRAINING=0: the (B, P)-block sequence)"""
        if not self.lowrank_chain_training or x.dtype != torch.float64 or only_last or amort is not None or self._poisson_column(si):
            return False
        if not isinstance(mlp, AmortizableMLP) or mlp.highway_mode != 0 or not mlp.use_permanent_parameters or not mlp.stages or mlp.linear is not None:
            return False
        last = mlp.stages[-1]
        if last["full"] or last["num_b"] == 0 or last["act"] or last["rank"] > _hip.LOWRANK_GF_MAX_RANK or len(mlp.sub_mlps) != 1:
            return False
        if layers[0].dimension > 8 or x.shape[0] == 0:
            return False
        for l in layers:
            c = l.c_struct()
            if l.has_extended_options:
                return False
            if not (c.num_kde == 10 and c.hh_iter <= 8 and c.nonlinear_stretch_type == _hip.GF_STRETCH_CLASSIC and c.width_mode == _hip.GF_WIDTH_SMOOTH
                    and not c.clamp_widths and c.fit_normalization and c.regulate_normalization):
                return False
        return True

    def _merge_candidate(self, x, data_summary, only_last, amort):
        """do the first two blocks of this log-prob step share a launch (csrc/merged_kernels.hip)?  They must be an unconditional broadcast g
        chain (<= 4 dimensions, classic layers) and a one-launch `f` block, in either order -- a structural property of the pdf and the switches
        below, decided once per key -- and the batch small enough for the merge to pay."""
        B = x.shape[0]
        if (x.dtype != torch.float32 or only_last or amort is not None or _hip.BINS_LOG is not None or type(data_summary) == list or B == 0
                or B > self.merge_max_rows or len(self.layer_list) < 2):
            return False
        key = (self.fuse_conditional_blocks, self.force_fused_manifold_blocks, data_summary is None, self.amortize_everything)
        hit = self._merge_ok.get(key)
        if hit is None:
            kinds = []
            for si in (0, 1):
                layers = list(self.layer_list[si])
                kind = self.pdf_defs_list[si][0]
                mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
                k = None
                if kind == "e" and mlp is None and not self.amortize_everything:
                    if (gfl.chain_supported(layers) and layers[0].dimension <= 4
                            and not any(l.nonlinear_stretch_type != "classic" or l.has_extended_options for l in layers)):
                        k = 0
                elif kind != "e" and len(layers) == 1:
                    mf = self._fusable_manifold_block(si, layers, False, None, x.dtype)
                    if mf is not None and mf[0] == "f":
                        k = 1
                kinds.append(k)
            hit = sorted(kinds, key=str) == [0, 1]
            self._merge_ok[key] = hit
        return hit

    def _fused_kernel_kind(self, n_rows):
        """which arithmetic of the register-resident fused block kernel (cond_split_kernels.hip): "split16" (f16 pairs, the default) or
        "split" (bf16 triples)"""
        if self.fused_matrix_arithmetic == "split_f16":
            return "split16"
        return "split"

    def _packed_w2(self, si, w2, b2, layer_array, n_layers, D, n_rows, kind=None):
        """(kind, packed split-bf16 image) of the block's output layer for the fused kernel chosen for this batch size, or None when the layer
        options are outside the kernels' set.  Rebuilt when the weights change: the key is the identity and in-place version of the MODULE's
        parameters (w2 / b2 may be casts of them made for this call -- fresh temporaries whose own version is always 0 and whose addresses
        the caching allocator hands out again), plus dtype, device and kernel kind."""
        kind = kind or self._fused_kernel_kind(n_rows)
        lin = self.mlp_predictors[si][2]
        key = (id(lin.weight), lin.weight._version, lin.weight.data_ptr(), id(lin.bias), lin.bias._version, lin.bias.data_ptr(), str(w2.dtype),
               str(w2.device), kind)
        hit = self._packed_cache.get((si, kind))
        if hit is not None and hit[0] == key:
            return hit[1]
        if _hip.cond_gf_packed_bytes(layer_array, n_layers, D, kind) < 0:
            packed = None                                                    # layer options outside the kernel's set
        else:
            packed = (kind, _hip.cond_gf_pack(w2, b2, layer_array, n_layers, D, kind))
        self._packed_cache[(si, kind)] = (key, packed)
        return packed

    def _block_params(self, si, data_summary, embeds, amort, counter):
        """extra_inputs row block of sub-pdf si, or None for permanent parameters (:936-993, 1420-1475)."""
        mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
        if mlp is not None:
            inp = self._mlp_input(si, data_summary, embeds)
            if not isinstance(mlp, HipLinearStack):
                inp = _hip.as_matrix(inp)
            if amort is not None:
                n = mlp.num_amortization_params
                out = mlp(inp, extra_inputs=amort[:, counter:counter + n])
                counter += n
            else:
                out = mlp(inp)
            if self._poisson_column(si):
                out = out[:, :-1]                           # the last output of the first MLP is log-lambda, not a flow parameter (:976-978)
            return out, counter
        if self.amortize_everything:
            assert amort is not None
            n = sum(l.get_total_param_num() for l in self.layer_list[si])
            if n > 0:
                out = amort[:, counter:counter + n]
                return out, counter + n
        return None, counter

    def _poisson_column(self, si):
        """does the amortisation MLP of sub-pdf si carry log-lambda as an extra last output column?"""
        return self.predict_log_normalization and self.join_poisson_and_pdf_description and si == 0 and self.conditional_input_dim is not None

    def _check_cond(self, x, conditional_input):
        if conditional_input is None:
            return
        if type(conditional_input) == list:
            assert len(self.conditional_input_dim) == len(conditional_input)
            for d, ci in zip(self.conditional_input_dim, conditional_input):
                assert d == ci.shape[1], "Inputs of conditional input vector do not match with pre-defined input_dims!"
                assert x.shape[0] == ci.shape[0], "Evaluating input x and condititional input shape must be similar!"
        else:
            assert x.shape[0] == conditional_input.shape[0], "Evaluating input x and condititional input shape must be similar!"

    def _status_ring(self, device):
        """a few pinned host slots + events for the deferred status read-back of the log-prob direction"""
        ring = getattr(self, "_ring", None)
        if ring is None or ring["device"] != device:
            ring = {"device": device, "host": [torch.zeros(_hip.JF_STATUS_WORDS, dtype=torch.int32).pin_memory() for _ in range(8)],
                    "event": [torch.cuda.Event() for _ in range(8)], "pending": [], "next": 0}
            self._ring = ring
        return ring

    def _defer_status(self, status):
        """check_status == "deferred": copy the status words to pinned host memory WITHOUT synchronising; they are examined when the copy has
        landed -- at the next call into this pdf or in flush_status() -- so a step does not end in a host-device round trip.  Any other
        truthy setting checks before returning."""
        if status is None or not self.check_status:
            return
        if self.check_status != "deferred":
            return self._report_status(status)
        ring = self._status_ring(status.device)
        if len(ring["pending"]) == len(ring["host"]):
            self._poll_status(block=True)
        i = ring["next"]
        ring["next"] = (i + 1) % len(ring["host"])
        ring["host"][i].copy_(status, non_blocking=True)
        ring["event"][i].record()
        ring["pending"].append(i)

    def _poll_status(self, block=False):
        ring = getattr(self, "_ring", None)
        while ring is not None and ring["pending"]:
            i = ring["pending"][0]
            if block:
                ring["event"][i].synchronize()
            elif not ring["event"][i].query():
                return
            ring["pending"].pop(0)
            self._report_status(ring["host"][i])

    def flush_status(self):
        """wait for every deferred status read-back and raise / warn exactly as an immediate check would have."""
        self._poll_status(block=True)
        for plan in self._step_plans.values():
            if plan:
                plan.flush()

    def _report_status(self, status):
        """kernel status words -> the reference's warnings / exceptions (bisection_n_newton.py:84-133, default.py:1516)."""
        if status is None or not self.check_status:
            return
        nonconv, nonfinite, oob, newton_steps = status.tolist()
        self.last_status_words = {"nonconverged": nonconv, "nonfinite": nonfinite, "out_of_range": oob, "newton_row_steps": newton_steps}
        if oob > 0:
            raise Exception("outside boundaries in rational-spline flow! (%d rows)" % oob)
        if nonfinite > 0:
            raise Exception("nonfinite values generated in %d rows .. this should never happen!" % nonfinite)
        if nonconv > 0:
            print(nonconv, " items did not converge in Newton iterations")

    # =========================================================================================== log-prob direction
    def _inverse_impl(self, x, log_det, data_summary, amortization_parameters, force_embedding_coordinates, force_intrinsic_coordinates,
                      only_last, want_base_logp, status, per_block=None):
        """`per_block` (optional list): receives the accumulated log_det after the coordinate transformation of each block (first
        len(layer_list) entries, only when a transformation is forced) and after each block's flow -- what the marginal entropies need."""
        _hip.require_device(x, log_det)
        if force_embedding_coordinates:
            assert x.shape[1] == self.total_target_dim_embedded, (x.shape[1], self.total_target_dim_embedded)
            x, log_det = self.transform_target_space(x, log_det, transform_from="embedding", transform_to="default", per_block=per_block)
        elif force_intrinsic_coordinates:
            assert x.shape[1] == self.total_target_dim_intrinsic
            x, log_det = self.transform_target_space(x, log_det, transform_from="intrinsic", transform_to="default", per_block=per_block)
        else:
            assert x.shape[1] == self.total_target_dim, (x.shape[1], self.total_target_dim)
        if amortization_parameters is not None:
            assert amortization_parameters.shape[1] == self.total_number_amortizable_params
        B = x.shape[0]
        base = torch.empty((B, self.total_base_dim), dtype=x.dtype, device=x.device)
        base_logp = None
        embeds = self._conditioning_rows(x, data_summary)
        lazy = embeds is None
        if lazy:
            embeds = []
        counter = 0
        # The blocks of the log-prob direction are independent given the targets (every MLP input is a function of x and the conditional
        # input alone, :946-962): each block returns its OWN log-det / base log-prob and one launch adds them up at the end
        # (_hip.combine_rows), instead of threading the running sums through the blocks as the reference does (:1020-1031).  A recorded plan
        # can then issue the blocks of a SMALL batch on side streams (csrc/plan.hip lanes): their launch / drain tails overlap.
        independent = per_block is None and not lazy and len(self.layer_list) > 1
        ld_parts = [] if log_det is None else [log_det]
        blp_parts = []
        rec = _hip._RECORDING
        lanes = rec is not None and independent and self.plan_lanes > 1 and B <= self.plan_lane_max_rows
        overlap = rec is not None and independent and not lanes and self.plan_overlap_blocks and B <= self.plan_overlap_max_rows
        # small batches: the launches of the first two blocks are captured by the library and issued as ONE grid (merged_kernels.hip)
        merge = (independent and not lanes and not overlap and self._merge_candidate(x, data_summary, only_last, amortization_parameters))
        if lanes:
            rec.fork()
        n_blocks = len(self.layer_list)
        folded_total = None
        if merge:
            _hip.merge_begin()
        try:
            return self._inverse_blocks(x, log_det, data_summary, amortization_parameters, force_embedding_coordinates, force_intrinsic_coordinates,
                                        only_last, want_base_logp, status, per_block, B, base, base_logp, embeds, lazy, counter, independent,
                                        ld_parts, blp_parts, rec, lanes, overlap, n_blocks, folded_total, merge)
        except BaseException:
            if merge:
                _hip.merge_abort()
            raise

    def _inverse_blocks(self, x, log_det, data_summary, amortization_parameters, force_embedding_coordinates, force_intrinsic_coordinates,
                        only_last, want_base_logp, status, per_block, B, base, base_logp, embeds, lazy, counter, independent, ld_parts, blp_parts,
                        rec, lanes, overlap, n_blocks, folded_total, merge):
        """the block loop of _inverse_impl (:998-1031)"""
        for si, block in enumerate(self.layer_list):
            if merge and si == 2:
                _hip.merge_end(x)                         # the first two blocks go out as one launch; the rest follows launch by launch
                merge = False
            if independent:
                if si > 0:
                    ld_parts.append(log_det)
                    if want_base_logp:
                        blp_parts.append(base_logp)
                log_det, base_logp = None, None
                if lanes:                                 # the last block stays on the caller's stream
                    rec.set_lane(0 if si == n_blocks - 1 else 1 + si % (self.plan_lanes - 1))
                if overlap and si == 1:                   # the blocks after the first may run beside their predecessors (no barrier bit)
                    rec.set_any_order(True)
            a, b = self.target_dim_indices[si]
            tgt = x[:, a:b]
            ba, bb = self.base_dim_indices[si]
            out_view = base[:, ba:bb]
            layers = list(block)
            kind = self.pdf_defs_list[si][0]
            fused = self._fusable_block(si, layers, only_last, amortization_parameters, x.dtype) if kind == "e" else None
            if fused is not None:
                # amortisation MLP + g layers in one launch: the per-sample parameter block never reaches HBM
                larr = _hip.gf_layer_array([l.c_struct() for l in layers])
                packed = None
                if self.fused_matrix_arithmetic != "f32" and fused[0].shape[0] <= 128:
                    packed = self._packed_w2(si, fused[2], fused[3], larr, len(layers), layers[0].dimension, x.shape[0])
                if packed is not None:
                    mlp_in = self._mlp_input(si, data_summary, embeds)
                    res = None
                    if independent and want_base_logp and si == n_blocks - 1 and not lanes and not overlap and self.fold_combine:
                        # the last block adds the earlier blocks' sums in its epilogue (list order, itself last: the bits of combine_rows) and
                        # writes log_prob = log_prob_base + log_det itself (:1110-1117): one launch fewer per step
                        res = _hip.cond_gf_chain_inv_split(mlp_in, fused[0], fused[1], packed[1], tgt, None, larr, len(layers), layers[0].dimension,
                                                           x_out=out_view, want_base_logp=True, status=status, kind=packed[0],
                                                           pre_ld=[t for t in ld_parts if t is not None], pre_blp=[t for t in blp_parts if t is not None])
                        if res is not None:
                            folded_total = res[3]
                    if res is None:
                        res = _hip.cond_gf_chain_inv_split(mlp_in, fused[0], fused[1], packed[1], tgt, log_det, larr,
                                                           len(layers), layers[0].dimension, x_out=out_view, base_logp_in=base_logp,
                                                           want_base_logp=want_base_logp, status=status, kind=packed[0])
                else:
                    res = _hip.cond_gf_chain_inv(_hip.as_matrix(self._mlp_input(si, data_summary, embeds)), *fused, tgt, log_det, larr, len(layers),
                                                 layers[0].dimension, x_out=out_view, base_logp_in=base_logp, want_base_logp=want_base_logp,
                                                 status=status)
                log_det = res[1]
                if want_base_logp:
                    base_logp = res[2]
                if lazy:
                    embeds.append(block[-1]._embedding_conditional_return(tgt))
                if per_block is not None:
                    per_block.append(log_det)
                continue
            lowrank = self._fusable_lowrank_block(si, layers, only_last, amortization_parameters, x) if kind == "e" else None
            if lowrank is not None:
                # low-rank AmortizableMLP + g layers in one launch: the parameter block is regenerated per lane from the row's rank-space vector
                res = _hip.amlp_gf_chain_inv(_hip.as_matrix(self._mlp_input(si, data_summary, embeds)), *lowrank, tgt, log_det,
                                             _hip.gf_layer_array([l.c_struct() for l in layers]), len(layers), layers[0].dimension,
                                             x_out=out_view, base_logp_in=base_logp, want_base_logp=want_base_logp, status=status)
                log_det = res[1]
                if want_base_logp:
                    base_logp = res[2]
                if lazy:
                    embeds.append(block[-1]._embedding_conditional_return(tgt))
                if per_block is not None:
                    per_block.append(log_det)
                continue
            mfused = self._fusable_manifold_block(si, layers, only_last, amortization_parameters, x.dtype) if kind != "e" else None
            if mfused is not None:
                # default amortisation MLP + the manifold chain in one launch: the parameter rows stay in LDS
                fam, ws = mfused
                structs = [l.c_struct() if fam == "r" else l.c_struct(1 if l.euclidean_to_sphere_as_first else 0) for l in layers]
                res = _hip.cond_mchain_inv(fam, _hip.as_matrix(self._mlp_input(si, data_summary, embeds)), *ws, tgt, log_det, structs, layers[0].dimension,
                                           x_out=out_view, base_logp_in=base_logp, want_base_logp=want_base_logp, status=status)
                if res is not None:
                    log_det = res[1]
                    if want_base_logp:
                        base_logp = res[2]
                    if lazy:
                        embeds.append(block[-1]._embedding_conditional_return(tgt))
                    if per_block is not None:
                        per_block.append(log_det)
                    continue
            extra, counter = self._block_params(si, data_summary, embeds, amortization_parameters, counter)
            if only_last:
                layers = layers[-1:]
            if kind == "e" and gfl.chain_supported(layers):
                if extra is None:
                    params = gfl.chain_permanent_row(layers, x)
                elif only_last:
                    params = extra[:, extra.shape[1] - layers[0].total_param_num:]
                else:
                    params = extra
                # a pdf that is (or, threading the sums, ends with) one plain g chain: the chain launch writes log_prob = base + log_det itself
                tot = (want_base_logp and not independent and si == n_blocks - 1 and per_block is None and self.fold_combine
                       and not force_embedding_coordinates and not force_intrinsic_coordinates)
                res = gfl.run_chain(layers, "inv", tgt, log_det, params, x_out=out_view, base_logp_in=base_logp,
                                    want_base_logp=want_base_logp, status=status, want_total=tot)
                log_det = res[1]
                if want_base_logp:
                    base_logp = res[2]
                if tot:
                    folded_total = res[3]
            elif _manifold_family(layers) is not None:
                params = extra
                if extra is not None and only_last:
                    params = extra[:, extra.shape[1] - layers[0].total_param_num:]
                pl = pb_ = None
                if (independent and want_base_logp and si == n_blocks - 1 and not lanes and not overlap and self.fold_combine
                        and _hip.BINS_LOG is None):
                    # the last block adds the earlier blocks' sums itself and writes log_prob (as the fused g block does): no combine launch
                    pl, pb_ = [t for t in ld_parts if t is not None], [t for t in blp_parts if t is not None]
                    if len(pl) > _hip.COND_GF_MAX_PRE or len(pb_) > _hip.COND_GF_MAX_PRE:
                        pl = pb_ = None
                res = _manifold_chain(_manifold_family(layers), layers, "inv", tgt, log_det, params, only_last and kind == "s", out_view,
                                      base_logp, want_base_logp, status, pre_ld=pl, pre_blp=pb_)
                log_det = res[1]
                if want_base_logp:
                    base_logp = res[2]
                if pl is not None:
                    folded_total = res[3]
            else:
                if log_det is None:
                    log_det = torch.zeros(B, dtype=x.dtype, device=x.device)
                cur = tgt
                used = 0
                for grp in reversed(_layer_groups(list(block))):       # tail-first parameter slices (:1002-1012)
                    n = sum(l.total_param_num for l in grp)
                    this = None
                    if extra is not None:
                        end = extra.shape[1] - used
                        this = extra[:, end - n:end]
                    if type(grp[0]) is gfl.gf_block and not only_last:
                        cur, log_det = gfl.run_chain(grp, "inv", cur, log_det, this if this is not None else gfl.chain_permanent_row(grp, x),
                                                     status=status)[:2]
                    else:
                        l = grp[-1]
                        if this is not None and len(grp) > 1:
                            this = this[:, n - l.total_param_num:]
                        kw = {}
                        if only_last and kind == "s":
                            kw["fix_euclidean_to_sphere_first"] = True
                        cur, log_det = l.inv_flow_mapping([cur, log_det], extra_inputs=this, **kw)[:2]
                    used += n
                    if only_last:
                        break
                out_view.copy_(cur)
                if want_base_logp:
                    base_logp = _hip.normal_logp(out_view, base_logp)
            if lazy:
                embeds.append(block[-1]._embedding_conditional_return(tgt))
            if per_block is not None:
                per_block.append(log_det)
        if merge:
            _hip.merge_end(x)                             # (a pdf of two blocks: both were captured)
        total = None
        if folded_total is not None:
            total = folded_total                          # log_det / base_logp are the totals already
        elif independent:
            ld_parts.append(log_det)
            if want_base_logp:
                blp_parts.append(base_logp)
            if lanes:
                rec.set_lane(0)
                rec.join()
            if overlap:
                rec.set_any_order(False)                  # the combine launch is ordered: it waits for every block
            log_det, base_logp, total = _hip.combine_rows([t for t in ld_parts if t is not None], [t for t in blp_parts if t is not None],
                                                          want_total=want_base_logp)
        elif want_base_logp:
            total = _hip.add_rows(base_logp, log_det)     # (:1110-1117)
        return base, log_det, base_logp, total

    def all_layer_inverse(self, x, log_det, data_summary, amortization_parameters=None, force_embedding_coordinates=False,
                          force_intrinsic_coordinates=False, only_last=False):
        """autoregressive backward mapping of all sub-manifold flows -> (base_pos, log_det)  (:879-1057)."""
        base, log_det, _, _ = self._inverse_impl(x, log_det, data_summary, amortization_parameters, force_embedding_coordinates,
                                                 force_intrinsic_coordinates, only_last, False, None)
        return base, log_det

    def forward(self, x, conditional_input=None, amortization_parameters=None, force_embedding_coordinates=False,
                force_intrinsic_coordinates=False, only_last=False):
        """log-probability at x -> (log_prob (B,), log_prob_base (B,), base_pos (B, D))  (:1059-1117)."""
        assert not self.use_as_passthrough_instead_of_pdf, "The module is only used as a passthrough of all layers, not as actually evaluating the pdf!"
        self._check_cond(x, conditional_input)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())
                                        or any(isinstance(c, torch.Tensor) and c.requires_grad for c in
                                               (conditional_input if type(conditional_input) == list else [conditional_input]))
                                        or (amortization_parameters is not None and amortization_parameters.requires_grad)):
            # training: the same kernels behind torch.autograd Functions (jammy_flows_amd/autograd.py)
            return self._forward_with_grad(x, conditional_input, amortization_parameters, force_embedding_coordinates,
                                           force_intrinsic_coordinates, only_last)
        with torch.no_grad():
            capturing = self._capture_status is not None
            if (self.use_step_plans and not capturing and amortization_parameters is None and not only_last and _hip.BINS_LOG is None
                    and (conditional_input is None or isinstance(conditional_input, torch.Tensor))):
                # the whole step re-issued from C by one call (jf_plan_launch): one plan per input signature, recorded at its first use
                plan = self._step_plan(x, conditional_input, force_embedding_coordinates, force_intrinsic_coordinates)
                if plan is not None:
                    return plan(x, conditional_input)
            if capturing:                                # inside graphed_forward() / a plan recording: a static status buffer, examined by the replaying side
                status = self._capture_status if _hip._RECORDING is not None else self._capture_status.zero_()
            else:
                self._poll_status()                      # surfaces problems of earlier calls whose status has arrived meanwhile
                status = _hip.new_status(x.device) if self.check_status else None
            base, log_det, log_pdf, total = self._inverse_impl(x, None, conditional_input, amortization_parameters, force_embedding_coordinates,
                                                               force_intrinsic_coordinates, only_last, True, status)
            if not capturing:
                self._defer_status(status)
        return total, log_pdf, base

    def planned_forward(self, x, conditional_input=None, **kwargs):
        """forward() for inputs of THIS signature (shape, strides, dtype, device) recorded once as a step plan (include/jammy_hip.h "step
        plans") -> callable(x, conditional_input=None) returning fresh (log_prob, log_prob_base, base) tensors.  A call is ONE ctypes call
        that issues every launch of the step from C; unlike graphed_forward() the inputs are read where they are (no copy into static
        buffers) and the outputs are new tensors.  The plan holds the addresses of the weights and of the caches built from them (packed
        images, flattened permanent rows); a call after a parameter update records again by itself (parameter version counters)."""
        return PlannedForward(self, x, conditional_input, kwargs)

    def pipelined_forward(self, x, conditional_input=None, depth=3, **kwargs):
        """forward() for a stream of independent batches of THIS signature -> PipelinedForward: submit(x, conditional_input) enqueues a step on
        one of `depth` alternating streams and returns a PendingStep (result() -> (log_prob, log_prob_base, base) once the caller's stream has
        been made to wait for it); drain() waits for all.  Keep `depth` steps in flight: the tail of one step overlaps the head of the next."""
        return PipelinedForward(self, x, conditional_input, depth, kwargs)

    def _step_plan(self, x, conditional_input, force_embedding_coordinates, force_intrinsic_coordinates):
        if not x.is_cuda or x.dim() != 2:
            return None
        key = (tuple(x.shape), x.stride(), x.dtype, x.device, x.data_ptr() % 16,
               None if conditional_input is None else (tuple(conditional_input.shape), conditional_input.stride(), conditional_input.dtype,
                                                       conditional_input.data_ptr() % 16),
               bool(force_embedding_coordinates), bool(force_intrinsic_coordinates),
               # the switches that choose kernels: a plan replays the choice made when it was recorded
               self.fuse_conditional_blocks, self.fused_matrix_arithmetic, self.fused_block_kernel, self.force_fused_manifold_blocks,
               self.plan_lanes, self.plan_overlap_blocks, self.fold_combine, self.merge_max_rows)
        plan = self._step_plans.get(key)
        if plan is None:
            if len(self._step_plans) >= 8:               # a few signatures per pdf (each plan keeps its intermediate buffers)
                self._step_plans.pop(next(iter(self._step_plans)))
            try:
                plan = PlannedForward(self, x, conditional_input, dict(force_embedding_coordinates=force_embedding_coordinates,
                                                                       force_intrinsic_coordinates=force_intrinsic_coordinates))
            except PlanNotApplicable:
                plan = False
            self._step_plans[key] = plan
        return plan or None

    def invalidate_packed_caches(self):
        """forget everything derived from the parameters: packed split images, int8 digit images, flattened permanent rows, step plans.  The
        caches follow the parameters' in-place version counters by themselves; writes that bypass them (`p.data.copy_(...)`, EMA weight
        swapping through `.data`) need this call."""
        self._packed_cache.clear()
        self._step_plans.clear()
        for m in self.modules():
            for attr in ("_i8_cache", "_packed", "_perm_row_cache", "_chain_cache"):
                if hasattr(m, attr):
                    try:
                        delattr(m, attr)
                    except AttributeError:
                        pass

    def graphed_forward(self, x, conditional_input=None, **kwargs):
        """forward() for inputs of THIS shape captured once in a HIP graph -> callable(x, conditional_input=None, check=True) returning
        (log_prob, log_prob_base, base).  A replay is one graph launch instead of 4 .. 12 kernel launches + their host-side preparation:
        what small batches (C1: 4096 rows, launch bound) and serving loops want.  The returned tensors are the graph's static output buffers
        (overwritten by the next call).  The graph reads the weights through the caches the eager path builds (packed split-bf16 images,
        flattened permanent rows): capture again after the parameters change."""
        return GraphedForward(self, x, conditional_input, kwargs)

    # =========================================================================================== log-prob direction, differentiable
    def _permanent_row_with_grad(self, layers, like):
        """the block's permanent parameters side by side in the extra_inputs layout, built with differentiable ops (torch.cat)"""
        parts = []
        for l in layers:
            if not hasattr(l, "_permanent_tensors"):       # a user-written layer keeps its own nn.Parameters: a placeholder of its width
                parts.append(torch.zeros(l.get_total_param_num(), dtype=like.dtype, device=like.device))
                continue
            parts += [t.reshape(-1).to(dtype=like.dtype) for t in l._permanent_tensors()]
        if not parts:
            return torch.zeros((1, 0), dtype=like.dtype, device=like.device)
        return torch.cat(parts).reshape(1, -1)

    def _forward_with_grad(self, x, conditional_input, amortization_parameters, force_embedding_coordinates, force_intrinsic_coordinates,
                           only_last, collect=None):
        """forward() with a torch.autograd graph: d log_prob / d (x, conditional_input, MLP weights, permanent layer parameters).
        Same launches as the inference path, wrapped in autograd Functions whose backward is a HIP launch (g chains, manifold chains) or
        rocBLAS GEMMs (dense layers) -- see jammy_flows_amd/autograd.py.
        collect (optional list): receives one dict per sub-pdf -- its target columns (a, b), its base coordinates y WITH their graph, and for
        pure g chains `cot`: v -> J_block^-T v, the co-vector carried through the block's layers by one launch (jf_gf_chain_inv_cot) -- what
        the implicit-function adjoint of sampling needs (_differentiable_sample)."""
        log_det0 = None
        if force_embedding_coordinates:          # the chart changes ahead of the block loop, with a graph (autograd.SphereEmbeddingFn)
            assert x.shape[1] == self.total_target_dim_embedded, (x.shape[1], self.total_target_dim_embedded)
            x, log_det0 = self.transform_target_space(x, None, transform_from="embedding", transform_to="default")
        elif force_intrinsic_coordinates:
            assert x.shape[1] == self.total_target_dim_intrinsic
            x, log_det0 = self.transform_target_space(x, None, transform_from="intrinsic", transform_to="default")
        amort = amortization_parameters
        if amort is not None:
            assert amort.shape[1] == self.total_number_amortizable_params
        counter = 0

        def last_only(layers, params, kind):
            """only_last (:1018): the block's last layer alone, with the tail of the block's parameter row (:1002-1012)"""
            if kind not in ("e", "s", "i"):
                raise Exception("Flow type ", kind, " does not supported *only_last*!")
            return layers[-1:], params[:, params.shape[1] - layers[-1].total_param_num:]

        def block_params(si, layers, inp, mlp):
            """(parameter block with grad, new counter): MLP output (own or per-sample weights), a slice of the amortisation block, or the
            permanent parameters (:936-993)"""
            nonlocal counter
            if mlp is not None:
                if amort is not None:
                    n = mlp.num_amortization_params
                    out = mlp(inp, extra_inputs=amort[:, counter:counter + n])
                    counter += n
                else:
                    out = mlp(inp)
                return out[:, :-1] if self._poisson_column(si) else out
            if self.amortize_everything:
                n = sum(l.get_total_param_num() for l in layers)
                out = amort[:, counter:counter + n]
                counter += n
                return out
            return self._permanent_row_with_grad(layers, x)
        _hip.require_device(x)
        self._poll_status()
        _hip.release_keepalive()                 # (tensors of the previous step that crossed streams: the streams have met since, _hip.KEEPALIVE)
        status = _hip.new_status(x.device) if self.check_status else None
        B = x.shape[0]
        log_det = log_det0
        base_logp = None
        bases = []
        embeds = []
        # independent blocks on side streams (self.train_streams): each block starts from zero sums, the sums are added after the loop
        n_blocks = len(self.layer_list)
        side = None
        if self.train_streams > 1 and n_blocks > 1 and collect is None and log_det0 is None:
            side = self._train_stream_objs.get(x.device)
            if side is None or len(side) != self.train_streams - 1:
                side = self._train_stream_objs[x.device] = [torch.cuda.Stream(device=x.device) for _ in range(self.train_streams - 1)]
                for st in side:
                    _hip.register_side_stream(st)          # (the backward nodes of these blocks announce their saved tensors, autograd._side_stream_safe)
        main_stream = torch.cuda.current_stream(x.device)
        ld_parts, blp_parts = [], []
        for si, block in enumerate(self.layer_list):
            a, b = self.target_dim_indices[si]
            tgt = x[:, a:b]
            layers = list(block)
            kind = self.pdf_defs_list[si][0]
            mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
            inp = None
            if mlp is not None:
                pieces = []
                if conditional_input is not None:
                    pieces.append(conditional_input[si] if type(conditional_input) == list else conditional_input)
                pieces += embeds
                if not pieces:
                    raise Exception("extra conditional input is empty but required for encoding!")
                inp = torch.cat(pieces, dim=1) if len(pieces) > 1 else pieces[0]
            stream_ctx = None
            if side is not None:
                log_det, base_logp = None, None
                if si < n_blocks - 1:                     # the last (usually largest) block stays on the caller's stream
                    st = side[si % len(side)]
                    st.wait_stream(main_stream)           # inputs (targets, embeddings, the cat above) were produced on the caller's stream
                    # ... and were allocated there: they must outlive this block's kernels (_hip.KEEPALIVE: held until the next gradient-mode
                    # forward call, by which time the caller's stream has waited for the side streams)
                    _hip.keep_alive(x, inp, amort, *embeds)
                    if conditional_input is not None:
                        _hip.keep_alive(*(conditional_input if type(conditional_input) == list else [conditional_input]))
                    stream_ctx = torch.cuda.stream(st)
                    stream_ctx.__enter__()
            try:
                if kind == "e" and gfl.chain_supported(layers):
                    D = layers[0].dimension
                    fused = self._fusable_block(si, layers, only_last, amort, x.dtype) if mlp is not None else None
                    if fused is not None:
                        larr = _hip.gf_layer_array([l.c_struct() for l in layers])
                        w1, b1, w2, b2 = mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias
                        packed = None
                        if self.fused_matrix_arithmetic != "f32" and w1.shape[0] <= 128:
                            packed = self._packed_w2(si, w2.detach(), b2.detach(), larr, len(layers), D, x.shape[0])
                        out, log_det, base_logp = autograd.CondBlockFn.apply(inp, w1, b1, w2, b2, tgt, log_det, base_logp, packed, larr, len(layers), D,
                                                                             status)
                    elif self._lowrank_chain_ok(si, layers, only_last, amort, x, mlp):
                        # low-rank last stage (float64): the chain regenerates its parameters from the rank-space vector, forward and backward
                        t2, u2, b2 = mlp.forward_to_last_rank(inp)
                        larr = _hip.gf_layer_array([l.c_struct() for l in layers])
                        out, log_det, base_logp = autograd.LowRankGfChainFn.apply(t2, u2, b2, tgt, log_det, base_logp, larr, len(layers), D, status)
                    else:
                        params = block_params(si, layers, inp, mlp)
                        used = layers
                        if only_last:
                            used, params = last_only(layers, params, kind)
                        larr = _hip.gf_layer_array([l.c_struct() for l in used])
                        out, log_det, base_logp = autograd.GfChainInvFn.apply(tgt, log_det, params, base_logp, larr, len(used), D, status)
                elif kind == "e":
                    # Euclidean block mixing 'g' runs with other layers ('t'): one launch per group, last group first (:1002-1012)
                    from ..layers.euclidean.multivariate_normal import mvn_block
                    params = block_params(si, layers, inp, mlp)
                    used = layers
                    if only_last:
                        used, params = last_only(layers, params, kind)
                    groups = _layer_groups(used)
                    out, c1 = tgt, params.shape[1]
                    for gi in range(len(groups) - 1, -1, -1):
                        grp = groups[gi]
                        n = sum(l.total_param_num for l in grp)
                        this = params[:, c1 - n:c1]
                        blp_in = base_logp if gi == 0 else None
                        if type(grp[0]) is gfl.gf_block:
                            larr = _hip.gf_layer_array([l.c_struct() for l in grp])
                            out, log_det, blp = autograd.GfChainInvFn.apply(out, log_det, this, blp_in, larr, len(grp), grp[0].dimension, status)
                        elif type(grp[0]) is mvn_block:
                            out, log_det, blp = autograd.TLayerInvFn.apply(out, log_det, this if n > 0 else None, blp_in, grp[0].c_struct(),
                                                                           grp[0].dimension, status)
                        else:
                            # a layer the library has no kernel for -- a user's euclidean_base subclass written in torch (layer_base.py:58-70): the
                            # layer's own operations carry the autograd graph; amortised parameters arrive as its extra_inputs slice
                            l = grp[-1]
                            own = None if (mlp is None and not self.amortize_everything) else (this[:, n - l.total_param_num:] if n > 0 else None)
                            ld_in = log_det if log_det is not None else torch.zeros(B, dtype=x.dtype, device=x.device)
                            out, log_det = l.inv_flow_mapping([out, ld_in], extra_inputs=own)[:2]
                            blp = None
                            if gi == 0:
                                blp = (-0.5 * out * out - 0.9189385332046727).sum(dim=1)
                                if blp_in is not None:
                                    blp = blp_in + blp
                        c1 -= n
                    base_logp = blp
                else:
                    params = block_params(si, layers, inp, mlp)
                    used = layers
                    if only_last:
                        used, params = last_only(layers, params, kind)
                    fam = _manifold_family(used)
                    groups = [used] if fam is not None else [[l] for l in used]          # mixed families (e.g. "mo"): one launch per layer
                    out, c1 = tgt, params.shape[1]
                    for gi in range(len(groups) - 1, -1, -1):                            # last layer first, parameters sliced tail-first (:1002-1012)
                        grp = groups[gi]
                        f = _manifold_family(grp)
                        if f is None:
                            raise NotImplementedError("gradients through %s layers are not implemented" % type(grp[0]).__name__)
                        n = sum(l.total_param_num for l in grp)
                        # (only_last on a sphere: the last layer also takes the sphere -> plane chart, fix_euclidean_to_sphere_first, :1018-1031)
                        structs = [l.c_struct() if f == "r" else l.c_struct(1 if (l.euclidean_to_sphere_as_first or (only_last and kind == "s")) else 0)
                                   for l in grp]
                        out, log_det, blp = autograd.MChainInvFn.apply(out, log_det, params[:, c1 - n:c1], base_logp if gi == 0 else None, f, structs,
                                                                       grp[0].dimension, status)
                        c1 -= n
                    base_logp = blp
            finally:
                if stream_ctx is not None:
                    stream_ctx.__exit__(None, None, None)
            if stream_ctx is not None:
                _hip.keep_alive(out, log_det, base_logp)                  # side-stream allocations, summed / concatenated on the caller's stream
            if side is not None:
                ld_parts.append(log_det)
                blp_parts.append(base_logp)
            bases.append(out)
            if collect is not None:
                cot = None
                if kind == "e" and gfl.chain_supported(layers) and not only_last and amort is None:
                    def cot(v, si=si, layers=layers, tgt=tgt.detach(), inp=None if inp is None else inp.detach(), mlp=mlp):
                        with torch.no_grad():
                            if mlp is not None:
                                params = mlp(inp)
                                params = params[:, :-1] if self._poisson_column(si) else params
                            else:
                                params = gfl.chain_permanent_row(layers, tgt)
                            return _hip.gf_chain_inv_cot(tgt, params, _hip.gf_layer_array([l.c_struct() for l in layers]), len(layers),
                                                         layers[0].dimension, v)
                collect.append({"a": a, "b": b, "y": out, "cot": cot, "coupled": mlp is not None and len(embeds) > 0})
            emb = block[-1]._embedding_conditional_return(tgt.detach()) if not tgt.requires_grad else autograd.embed(tgt, kind, layers[-1])
            embeds.append(emb)
        if side is not None:
            for st in side:
                main_stream.wait_stream(st)               # every block's outputs are complete before the caller's stream adds them up
            if autograd.COMBINE_ROWS_FN and all(t is not None and t.dim() == 1 for t in ld_parts + blp_parts):
                # the blocks' sums and the total in one launch (autograd.CombineRowsFn)
                total, base_logp = autograd.CombineRowsFn.apply(len(ld_parts), *ld_parts, *blp_parts)
                base = torch.cat(bases, dim=1) if len(bases) > 1 else bases[0]
                self._defer_status(status)
                return total, base_logp, base
            log_det = ld_parts[0]
            for t in ld_parts[1:]:
                log_det = log_det + t
            base_logp = blp_parts[0]
            for t in blp_parts[1:]:
                base_logp = base_logp + t
        base = torch.cat(bases, dim=1) if len(bases) > 1 else bases[0]
        total = base_logp + log_det
        self._defer_status(status)
        return total, base_logp, base

    def log_prob(self, x, conditional_input=None, **kwargs):
        """convenience: forward(...)[0]  (the reference has no such method)."""
        return self.forward(x, conditional_input=conditional_input, **kwargs)[0]

    # =========================================================================================== sampling direction
    def all_layer_forward(self, x, log_det, data_summary, amortization_parameters=None, force_embedding_coordinates=False,
                          force_intrinsic_coordinates=False, only_last=False, status=None, per_block=None):
        """autoregressive forward mapping base -> target -> (x, log_det)  (:1373-1531).  `per_block` (optional list) receives the accumulated
        log_det after each block's flow and then after each block's coordinate transformation (if one is forced)."""
        _hip.require_device(x, log_det)
        if amortization_parameters is not None:
            assert amortization_parameters.shape[1] == self.total_number_amortizable_params
        else:
            assert not self.amortize_everything
        B = x.shape[0]
        out = torch.empty((B, self.total_target_dim), dtype=x.dtype, device=x.device)
        # the MLP input rows cat[conditional_input, embed(x_0), ...] (:1440-1456) as SEGMENTS of the output buffer, which the blocks fill one
        # after the other: block si reads the columns of the blocks before it where they are (the consumers of _mlp_input with segment
        # support), as in the log-prob direction -- no torch.cat, no embedding launch per sphere block
        embeds = self._conditioning_rows(out, data_summary)
        lazy = embeds is None
        if lazy:
            embeds = []
        counter = 0
        for si, block in enumerate(self.layer_list):
            kind = self.pdf_defs_list[si][0]
            layers = list(block)
            fused = self._fusable_block(si, layers, only_last, amortization_parameters, x.dtype) if kind == "e" else None
            if fused is not None and self.fused_matrix_arithmetic != "f32" and fused[0].shape[0] <= 128:
                # amortisation MLP + the g layers' solves in one launch, parameters regulated once in the MFMA result registers
                larr = _hip.gf_layer_array([l.c_struct() for l in layers])
                packed = self._packed_w2(si, fused[2], fused[3], larr, len(layers), layers[0].dimension, 0,
                                         kind="split16" if self.fused_matrix_arithmetic == "split_f16" else "split")
                if packed is not None:
                    ba, bb = self.base_dim_indices[si]
                    a, b = self.target_dim_indices[si]
                    _, log_det = _hip.cond_gf_chain_fwd_split(self._mlp_input(si, data_summary, embeds), fused[0], fused[1], packed[1], x[:, ba:bb],
                                                              log_det, larr, len(layers), layers[0].dimension, x_out=out[:, a:b], status=status,
                                                              kind=packed[0])
                    if lazy:
                        embeds.append(block[-1]._embedding_conditional_return(out[:, a:b]))
                    if per_block is not None:
                        per_block.append(log_det)
                    continue
            lowrank = self._fusable_lowrank_block(si, layers, only_last, amortization_parameters, x) if kind == "e" else None
            if lowrank is not None:
                # low-rank AmortizableMLP + the g layers' solves in one launch (float64, ranks <= 8): no (B, N) parameter block in HBM
                ba, bb = self.base_dim_indices[si]
                a, b = self.target_dim_indices[si]
                res = _hip.amlp_gf_chain_fwd(_hip.as_matrix(self._mlp_input(si, data_summary, embeds)), *lowrank, x[:, ba:bb], log_det,
                                             _hip.gf_layer_array([l.c_struct() for l in layers]), len(layers), layers[0].dimension,
                                             x_out=out[:, a:b], status=status)
                if res is not None:
                    log_det = res[1]
                    if lazy:
                        embeds.append(block[-1]._embedding_conditional_return(out[:, a:b]))
                    if per_block is not None:
                        per_block.append(log_det)
                    continue
            mfused = self._fusable_manifold_block(si, layers, only_last, amortization_parameters, x.dtype) if kind != "e" else None
            if mfused is not None:
                # default amortisation MLP + the manifold chain forwards in one launch: the parameter rows stay in LDS
                fam, ws = mfused
                structs = [l.c_struct() if fam == "r" else l.c_struct(1 if l.euclidean_to_sphere_as_first else 0) for l in layers]
                ba, bb = self.base_dim_indices[si]
                a, b = self.target_dim_indices[si]
                res = _hip.cond_mchain_fwd(fam, _hip.as_matrix(self._mlp_input(si, data_summary, embeds)), *ws, x[:, ba:bb], log_det, structs,
                                           layers[0].dimension,
                                           x_out=out[:, a:b], status=status)
                if res is not None:
                    log_det = res[1]
                    if lazy:
                        embeds.append(block[-1]._embedding_conditional_return(out[:, a:b]))
                    if per_block is not None:
                        per_block.append(log_det)
                    continue
            extra, counter = self._block_params(si, data_summary, embeds, amortization_parameters, counter)
            ba, bb = self.base_dim_indices[si]
            cur = x[:, ba:bb]
            a, b = self.target_dim_indices[si]
            out_view = out[:, a:b]
            if only_last:
                if kind not in "es":
                    raise Exception("Flow type ", kind, " does not supported *only_last*!")
                layers = layers[-1:]
            if kind == "e" and gfl.chain_supported(layers):
                if extra is None:
                    params = gfl.chain_permanent_row(layers, x)
                elif only_last:
                    params = extra[:, extra.shape[1] - layers[0].total_param_num:]
                else:
                    params = extra
                _, log_det = gfl.run_chain(layers, "fwd", cur, log_det, params, x_out=out_view, status=status)
            elif _manifold_family(layers) is not None:
                params = extra
                if extra is not None and only_last:
                    params = extra[:, extra.shape[1] - layers[0].total_param_num:]
                _, log_det = _manifold_chain(_manifold_family(layers), layers, "fwd", cur, log_det, params, only_last and kind == "s", out_view,
                                             None, False, status)
            else:
                if log_det is None:
                    log_det = torch.zeros(B, dtype=x.dtype, device=x.device)
                c = 0
                groups = [[l] for l in block] if only_last else _layer_groups(list(block))
                for gi, grp in enumerate(groups):
                    n = sum(l.total_param_num for l in grp)
                    this = None if extra is None else extra[:, c:c + n]
                    c += n
                    if only_last and gi < len(groups) - 1:
                        continue
                    if type(grp[0]) is gfl.gf_block and not only_last:
                        _, log_det = gfl.run_chain(grp, "fwd", cur, log_det, this if this is not None else gfl.chain_permanent_row(grp, x),
                                                   status=status)
                        cur = _
                    else:
                        kw = {}
                        if only_last and kind == "s":
                            kw["fix_euclidean_to_sphere_first"] = True
                        cur, log_det = grp[0].flow_mapping([cur, log_det], extra_inputs=this, **kw)[:2]
                out_view.copy_(cur)
            if lazy:
                embeds.append(block[-1]._embedding_conditional_return(out_view))
            if per_block is not None:
                per_block.append(log_det)
        x_new = out
        if force_embedding_coordinates:
            x_new, log_det = self.transform_target_space(x_new, log_det, transform_from="default", transform_to="embedding", per_block=per_block)
        elif force_intrinsic_coordinates:
            x_new, log_det = self.transform_target_space(x_new, log_det, transform_from="default", transform_to="intrinsic", per_block=per_block)
        return x_new, log_det

    def _obtain_sample(self, conditional_input=None, predefined_target_input=None, samplesize=1, seed=None, amortization_parameters=None,
                       force_embedding_coordinates=False, force_intrinsic_coordinates=False, failsafe_crosscheck_tolerance=None, dtype=None,
                       device=None, only_last=False):
        """base noise (drawn with numpy on the host like the reference does, or injected) -> (x, base, log_prob, log_prob_base)  (:1533-1707)."""
        if failsafe_crosscheck_tolerance:
            raise NotImplementedError("failsafe_crosscheck_tolerance (recheck_sampling) is outside the MI355X hot path")
        used = samplesize
        if self.amortize_everything:
            assert amortization_parameters is not None
            dev, dt, used = amortization_parameters.device, amortization_parameters.dtype, amortization_parameters.shape[0]
        elif conditional_input is not None:
            ci = conditional_input[0] if type(conditional_input) == list else conditional_input
            used, dt, dev = ci.shape[0], ci.dtype, ci.device
        else:
            dt, dev = self.obtain_current_dtype_n_device()
            if device is not None:
                dev = device
            if dtype is not None:
                dt = dtype
        assert dt is not None and dev is not None, "DType and/or device is None: pass dtype and device as keyword arguments"
        if predefined_target_input is not None:
            z = predefined_target_input
            if conditional_input is not None:
                ci = conditional_input[0] if type(conditional_input) == list else conditional_input
                assert z.shape[0] == ci.shape[0] and z.dtype == ci.dtype and z.device == ci.device
            base_ret = 0.0
        else:
            if seed is not None:                 # reproducible draws: the reference's own generator and call (:1634-1657), on the host
                numpy.random.seed(seed)
                z = torch.from_numpy(numpy.random.normal(size=(used, self.total_base_dim))).type(dt).to(dev)
            else:                                # unseeded: drawn on the device (the host generator + copy cost 30x the sampling kernels)
                z = torch.randn((used, self.total_base_dim), dtype=dt, device=dev)
            base_ret = z
        _hip.require_device(z)
        status = _hip.new_status(z.device) if self.check_status else None
        log_gauss = _hip.normal_logp(z, None)
        x, log_det = self.all_layer_forward(z, None, conditional_input, amortization_parameters=amortization_parameters,
                                            force_embedding_coordinates=force_embedding_coordinates,
                                            force_intrinsic_coordinates=force_intrinsic_coordinates, only_last=only_last, status=status)
        self._report_status(status)
        return x, base_ret, log_gauss - log_det, log_gauss          # (one launch; equal to -log_det + log_gauss bit for bit)

    def _differentiable_sample(self, conditional_input=None, predefined_target_input=None, samplesize=1, seed=None, amortization_parameters=None,
                               force_embedding_coordinates=False, force_intrinsic_coordinates=False, dtype=None, device=None, only_last=False):
        """samples that carry gradients with respect to the pdf's parameters and the conditional input (the reference
        differentiates through its Newton iterations, bisection_n_newton.py:74-93).

        Here the sample x* = F_theta^-1(z) comes from the sampling kernels without a graph, and the gradient from the implicit-function
        theorem, written as ONE differentiable Newton correction at the solution:

            x = x* - J^-1 (F_theta(x*) - z),        J = dF/dx at x* (held constant)

        Its value is x* (the bracket vanishes to solver precision), its derivative -J^-1 dF/dtheta is exactly d x*/d theta.  F_theta(x*) is the
        log-prob direction with a graph (the backward kernels of autograd.py).  J is never formed: the backward of the correction needs
        lambda = J^-T g, and J is block lower triangular over the autoregressive sub-pdfs, so lambda comes from ONE back-substitution -- per
        block a co-vector launch through its g layers (J_block^-T: the layers' reflections and a division by each stage's derivative) and one
        input-gradient pass for the coupling to the earlier blocks (autograd.InverseJacobianFn); blocks without that launch (manifold layers,
        't' layers) use their small dense Jacobian block.  log_prob gets the matching first-order term:
        log p_theta(x*) + <grad_x log p, x - x*>."""
        if only_last:
            raise NotImplementedError("differentiable sampling works on the full pdf (only_last is not supported)")
        with torch.no_grad():
            x_star, base_ret, _, logp_base = self._obtain_sample(conditional_input=conditional_input, predefined_target_input=predefined_target_input,
                                                                 samplesize=samplesize, seed=seed, amortization_parameters=amortization_parameters,
                                                                 force_intrinsic_coordinates=force_intrinsic_coordinates, dtype=dtype, device=device)
            z = predefined_target_input if predefined_target_input is not None else base_ret
        x0 = x_star.detach().clone().requires_grad_(True)
        blocks = []
        logp_x, _, y = self._forward_with_grad(x0, conditional_input, amortization_parameters, False, False, False, collect=blocks)
        D = y.shape[1]
        assert D == x0.shape[1], "differentiable sampling needs a square Jacobian (intrinsic coordinates)"
        (g_logp,) = torch.autograd.grad(logp_x.sum(), x0, retain_graph=True)
        # delta = J^-1 (y - z) is zero to solver precision; its BACKWARD is lambda = J^-T g, found by back-substitution over the autoregressive
        # blocks (autograd.InverseJacobianFn), after which autograd walks the graph of y once with upstream lambda: ~3 backward passes in all
        # (rounds 2-3: D + 2 -- one per row of a dense Jacobian -- and a batched LAPACK solve)
        delta = autograd.InverseJacobianFn.apply(y - z.detach(), x0, blocks)
        x = x0.detach() - delta
        logp = logp_x - (g_logp * delta).sum(dim=1)
        if force_embedding_coordinates:          # the chart change behind the flow, with its graph (:1508-1531): log p picks up its log-det
            x, ld_emb = self.transform_target_space(x, torch.zeros_like(logp), transform_from="default", transform_to="embedding")
            logp = logp - ld_emb
        return x, base_ret, logp, logp_base

    def obtain_flow_param_structure(self, conditional_input=None, predefined_target_input=None, seed=None, dtype=None, device=None):
        """names and values of every layer's parameters for the given conditional input, walking the sampling direction layer by layer
        (:1119-1287; debugging / plotting).  Keys: "<sub-pdf index>_<flow def>.<layer index>" -> ordered dict of named tensors."""
        import collections
        p0 = next(iter(self.parameters()), None)
        data_type = dtype if dtype is not None else (p0.dtype if p0 is not None else torch.float64)
        used_device = device if device is not None else (p0.device if p0 is not None else torch.device("cuda"))
        n = 1
        if conditional_input is not None:
            first = conditional_input[0] if type(conditional_input) == list else conditional_input
            n, data_type, used_device = first.shape[0], first.dtype, first.device
        if predefined_target_input is not None:
            x = predefined_target_input
            if conditional_input is None:
                n, data_type, used_device = x.shape[0], x.dtype, x.device
        else:
            if seed is not None:
                numpy.random.seed(seed)
            x = torch.from_numpy(numpy.random.normal(size=(n, self.total_base_dim))).to(dtype=data_type, device=used_device)
        log_det = torch.zeros(n, dtype=data_type, device=used_device)
        structure = collections.OrderedDict()
        embeds = []
        with torch.no_grad():
            for si, block in enumerate(self.layer_list):
                mlp = self.mlp_predictors[si] if len(self.mlp_predictors) > si else None
                extra = None
                if mlp is not None:
                    pieces = []
                    if conditional_input is not None:
                        pieces.append(conditional_input[si] if type(conditional_input) == list else conditional_input)
                    pieces += embeds
                    if not pieces:
                        raise Exception("SAMPLE: extra conditional input is empty but required for encoding!")
                    extra = mlp(torch.cat(pieces, dim=1) if len(pieces) > 1 else pieces[0])
                    if self._poisson_column(si):
                        extra = extra[:, :-1]                   # (:1257-1260)
                a, b = self.base_dim_indices[si]
                cur = x[:, a:b]
                c = 0
                for li, layer in enumerate(block):
                    this = None if extra is None else extra[:, c:c + layer.total_param_num]
                    d = collections.OrderedDict()
                    layer.obtain_layer_param_structure(d, extra_inputs=this, previous_x=cur)
                    structure[("%.3d" % si) + "_" + self.flow_defs_list[si] + ".%.3d" % li] = d
                    cur, log_det = layer.flow_mapping([cur, log_det], extra_inputs=this)[:2]
                    c += layer.total_param_num
                embeds.append(block[-1]._embedding_conditional_return(cur))
        return structure

    def sample(self, conditional_input=None, samplesize=1, seed=None, allow_gradients=False, amortization_parameters=None,
               force_embedding_coordinates=False, force_intrinsic_coordinates=False, failsafe_crosscheck_tolerance=None, dtype=None,
               device=None, only_last=False):
        """draw samples -> (x, base, log_prob, log_prob_base)  (:1300-1371)."""
        assert not self.use_as_passthrough_instead_of_pdf
        if allow_gradients and torch.is_grad_enabled():
            return self._differentiable_sample(conditional_input=conditional_input, samplesize=samplesize, seed=seed,
                                               amortization_parameters=amortization_parameters,
                                               force_embedding_coordinates=force_embedding_coordinates,
                                               force_intrinsic_coordinates=force_intrinsic_coordinates, device=device, dtype=dtype,
                                               only_last=only_last)
        with torch.no_grad():
            return self._obtain_sample(conditional_input=conditional_input, seed=seed, samplesize=samplesize,
                                       amortization_parameters=amortization_parameters,
                                       force_embedding_coordinates=force_embedding_coordinates,
                                       force_intrinsic_coordinates=force_intrinsic_coordinates,
                                       failsafe_crosscheck_tolerance=failsafe_crosscheck_tolerance, device=device, dtype=dtype,
                                       only_last=only_last)

    # =========================================================================================== analysis reductions
    def approximate_coverage(self, target_x, conditional_input=None, amortization_parameters=None, force_embedding_coordinates=False,
                             force_intrinsic_coordinates=False, num_percentile_points=100, sub_manifolds=[-1]):
        """approximate coverage through the base distribution: 2 (log p_base(0) - log p_base(z)) is chi^2 distributed for a calibrated pdf
        (:1954-2022, helper_fns/coverage.py:45-65).  Same return dictionary as the reference ("expected", "true", "logprob_diffs",
        "chi2_cdf_evals", keyed "total" / sub-manifold index).  The forward pass and the counting run on the device (jf_coverage_histogram):
        only the histogram -- and the per-row arrays the reference's API hands back -- go to the host."""
        from scipy import stats
        expected = numpy.linspace(0, 1.0, num_percentile_points)
        ret = {"true": {}, "logprob_diffs": {}, "chi2_cdf_evals": {}, "expected": expected}
        with torch.no_grad():
            _, logp_base, base = self.forward(target_x, conditional_input=conditional_input, amortization_parameters=amortization_parameters,
                                              force_embedding_coordinates=force_embedding_coordinates,
                                              force_intrinsic_coordinates=force_intrinsic_coordinates)

            def one(lpb, dim):
                thr = torch.from_numpy(stats.chi2.ppf(expected, df=dim)).to(device=lpb.device)
                counts, twice = _hip.coverage_histogram(lpb, -(dim / 2.0) * numpy.log(2 * numpy.pi), thr)
                twice = twice.double().cpu().numpy()
                return counts.astype(numpy.float64) / float(lpb.shape[0]), twice, stats.chi2.cdf(twice, df=dim)

            for sm in sub_manifolds:
                if sm == -1:
                    key, (t, d, c) = "total", one(logp_base, self.total_base_dim)
                else:
                    assert 0 <= sm < len(self.pdf_defs_list), "Sub manifold index %d is invalid" % sm
                    a, b = self.target_dim_indices_intrinsic[sm]            # the reference indexes base_points with the intrinsic target ranges
                    key, (t, d, c) = int(sm), one(_hip.normal_logp(base[:, a:b]), self.target_dims_intrinsic[sm])
                ret["true"][key], ret["logprob_diffs"][key], ret["chi2_cdf_evals"][key] = t, d, c
        return ret

    def entropy(self, sub_manifolds=[-1], conditional_input=None, force_embedding_coordinates=True, force_intrinsic_coordinates=False,
                samplesize=100, failsafe_crosscheck_tolerance=None, dtype=None, device=None, predefined_base=None):
        """Monte-Carlo entropy of the pdf and of the marginal pdfs of single sub-manifolds (:2263-2454): -mean log p over `samplesize`
        samples per conditional input; for the marginal of sub-manifold k > 0 the conditional density p(x_k | x_<k) of each of the S
        samples is averaged over the S draws of x_<k (log-mean-exp over an S x S evaluation).  Sampling, the S^2-row log-prob pass and the
        sample means (jf_segment_reduce) all stay on the device.  `predefined_base` injects the standard-normal base samples (tests)."""
        if failsafe_crosscheck_tolerance:
            raise NotImplementedError("failsafe_crosscheck_tolerance (recheck_sampling) is outside the MI355X hot path")
        dt, dev = self.obtain_current_dtype_n_device()
        dev = device if device is not None else dev
        dt = dtype if dtype is not None else dt
        S = samplesize
        data_summary, batch = None, 1
        if conditional_input is not None:
            assert self.conditional_input_dim is not None
            if type(conditional_input) == list:
                dt, dev, batch = conditional_input[0].dtype, conditional_input[0].device, conditional_input[0].shape[0]
                data_summary = [ci.repeat_interleave(S, dim=0) for ci in conditional_input]
            else:
                dt, dev, batch = conditional_input.dtype, conditional_input.device, conditional_input.shape[0]
                data_summary = conditional_input.repeat_interleave(S, dim=0)
        else:
            assert self.conditional_input_dim is None, "We require conditional input, since this is a conditional PDF."
        for sm in sub_manifolds:
            assert sm == -1 or 0 <= sm < len(self.layer_list)
        nsub = len(self.layer_list)
        out = {}
        with torch.no_grad():
            z = predefined_base if predefined_base is not None else torch.randn((S * batch, self.total_base_dim), dtype=dt, device=dev)
            assert z.shape == (S * batch, self.total_base_dim)
            status = _hip.new_status(z.device) if self.check_status else None
            marks = []
            targets, log_det = self.all_layer_forward(z, None, data_summary, force_embedding_coordinates=force_embedding_coordinates,
                                                      force_intrinsic_coordinates=force_intrinsic_coordinates, status=status, per_block=marks)
            self._report_status(status)
            forced = force_embedding_coordinates or force_intrinsic_coordinates

            def increments(acc, start):
                """accumulated log-dets (None = nothing added yet) -> per-block increments"""
                res, prev = [], start
                for m in acc:
                    cur = 0.0 if m is None else m
                    res.append(cur - (0.0 if prev is None else prev))
                    prev = m
                return res
            # sampling direction: flow marks first, then (if forced) the marks of the coordinate transformation
            flow_inc = increments(marks[:nsub], None)
            trans_inc = increments(marks[nsub:], marks[nsub - 1]) if forced else [0.0] * nsub
            if -1 in sub_manifolds:
                out["total"] = _hip.segment_reduce(_hip.normal_logp(z) - log_det, S, "neg_mean")
            for sm in sub_manifolds:
                if sm == -1:
                    continue
                ba, bb = self.base_dim_indices[sm]
                if sm == 0:
                    out[0] = _hip.segment_reduce(_hip.normal_logp(z[:, ba:bb]) - (flow_inc[0] + trans_inc[0]), S, "neg_mean")
                    continue
                dims = self.target_dims_embedded if force_embedding_coordinates else (
                    self.target_dims_intrinsic if force_intrinsic_coordinates else self.target_dims)
                first = sum(dims[:sm])
                w = dims[sm]
                # rows (g, i, j): x_<k of sample j, x_k of sample i of conditional input g; later sub-manifolds filled with ones (:2413-2416)
                tg = targets.reshape(batch, S, -1)
                prev = tg[:, None, :, :first].expand(batch, S, S, first)
                fin = tg[:, :, None, first:first + w].expand(batch, S, S, w)
                fill = torch.ones((batch, S, S, targets.shape[1] - first - w), dtype=targets.dtype, device=targets.device)
                filled = torch.cat([prev, fin, fill], dim=3).reshape(batch * S * S, -1)
                ds2 = None
                if data_summary is not None:
                    ds2 = ([d.repeat_interleave(S, dim=0) for d in data_summary] if type(data_summary) == list
                           else data_summary.repeat_interleave(S, dim=0))
                marks2 = []
                base2, _, _, _ = self._inverse_impl(filled, None, ds2, None, force_embedding_coordinates, force_intrinsic_coordinates, False, False, None,
                                                 per_block=marks2)
                # log-prob direction: (if forced) transformation marks first, then the flow marks continuing from their total
                if forced:
                    t_inc = increments(marks2[:nsub], None)
                    f_inc = increments(marks2[nsub:], marks2[nsub - 1])
                else:
                    t_inc, f_inc = [0.0] * nsub, increments(marks2, None)
                lp = _hip.normal_logp(base2[:, ba:bb]) + f_inc[sm] + t_inc[sm]
                out[sm] = _hip.segment_reduce(_hip.segment_reduce(lp, S, "logmeanexp"), S, "neg_mean")
        return out

    # =========================================================================================== coordinate systems
    def transform_target_into_returnable_params(self, target):
        return self.transform_target_space(target)[0]

    def transform_target_space(self, target, log_det=0, transform_from="default", transform_to="embedding", per_block=None):
        """default / intrinsic / embedding coordinates of the target tensor (:1737-1813)."""
        new_target = target.unsqueeze(0) if target.dim() == 1 else target
        dims = {"default": self.target_dims, "intrinsic": self.target_dims_intrinsic, "embedding": self.target_dims_embedded}
        totals = {"default": self.total_target_dim, "intrinsic": self.total_target_dim_intrinsic, "embedding": self.total_target_dim_embedded}
        if transform_from not in dims or transform_to not in dims:
            raise Exception("Unknown transformation space! Allowed: default/intrinsic/embedding")
        assert new_target.shape[1] == totals[transform_from]
        vals = []
        c = 0
        for si, block in enumerate(self.layer_list):
            n = dims[transform_from][si]
            t, log_det = block[-1].transform_target_space(new_target[:, c:c + n], log_det=log_det, transform_from=transform_from,
                                                          transform_to=transform_to)
            if per_block is not None:
                per_block.append(log_det)
            vals.append(t)
            c += n
        res = torch.cat(vals, dim=1) if len(vals) > 1 else vals[0]
        assert res.shape[1] == totals[transform_to]
        if target.dim() == 1:
            res = res.squeeze(0)
        return res, log_det


class PlanNotApplicable(RuntimeError):
    """the step of this configuration cannot be replayed from a plan (it leaves the library between launches)"""


class _RecordingPass(torch.utils._python_dispatch.TorchDispatchMode):
    """active while a step is recorded.  Two jobs, both through the dispatcher (every ATen call of this thread passes here):
    (1) every tensor the pass creates stays alive until the pass is over, so the caching allocator cannot hand the memory of a dead
        intermediate to a later allocation OF THE SAME PASS -- the plan tells buffers apart by address range, and an output tensor carved out of
        a dead, larger intermediate would capture the pointers to that intermediate;
    (2) any ATen operation on device tensors other than allocations and views is work the plan cannot replay (it ran once, now): noted in
        `foreign`, the configuration then stays on the eager path."""
    ALLOWED = {"aten.empty.memory_format", "aten.empty_strided.default", "aten.empty_like.default", "aten.new_empty.default",
               "aten.slice.Tensor", "aten.select.int", "aten.as_strided.default", "aten.view.default", "aten._unsafe_view.default",
               "aten.unsqueeze.default", "aten.squeeze.dim", "aten.squeeze.default", "aten.expand.default", "aten.t.default",
               "aten.transpose.int", "aten.detach.default", "aten.alias.default", "aten.narrow.default", "aten.permute.default",
               "aten.lift_fresh.default", "aten.is_pinned.default", "aten._reshape_alias.default"}

    def __init__(self):
        super().__init__()
        self.keep, self.foreign = [], []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        flat = [t for t in torch.utils._pytree.tree_leaves((args, kwargs, out)) if isinstance(t, torch.Tensor)]
        if any(t.is_cuda for t in flat):
            self.keep.append(out)
            if str(func) not in self.ALLOWED:
                self.foreign.append(str(func))
        return out


# Memory pools of dead plans.  torch frees a MemPool's cached blocks in its destructor (emptyCache), which ASSERTS that no allocate-to-pool
# context is active: a cyclic-GC pass that happens to collect an old PlannedForward while another one is being recorded aborted the process.
# A dying plan therefore parks its pool here; the pools are released at a safe point (before the next recording, or on request).
_RETIRED_POOLS = []


def release_plan_memory():
    """free the private memory pools of step plans that are no longer referenced"""
    while _RETIRED_POOLS:
        _RETIRED_POOLS.pop()


# every nn.Module.register_parameter call of the process (a replaced Parameter object anywhere): PlannedForward._param_key walks its pdf's
# parameters again when this moved
_PARAM_REGISTRATIONS = [0]


def _count_parameter_registration(module, name, param):
    _PARAM_REGISTRATIONS[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_count_parameter_registration)


class PlannedForward:
    """pdf.forward recorded as a step plan for one input signature (see pdf.planned_forward)."""

    def __del__(self):
        try:
            _RETIRED_POOLS.append((getattr(self, "out_like", None), getattr(self, "pool", None)))
        except Exception:           # noqa: BLE001 -- interpreter shutdown
            pass

    def __init__(self, pdf, x, conditional_input, kwargs):
        dev = _hip.require_device(x, conditional_input if isinstance(conditional_input, torch.Tensor) else None)
        if x.dim() != 2 or (conditional_input is not None and not isinstance(conditional_input, torch.Tensor)):
            raise PlanNotApplicable("step plans take one (B, D) target tensor and at most one conditional-input tensor")
        if x.shape[0] == 0:
            raise PlanNotApplicable("empty batch")
        if conditional_input is not None:
            def span(t):
                return t.data_ptr(), t.data_ptr() + (sum((n - 1) * st for n, st in zip(t.shape, t.stride())) + 1) * t.element_size()
            (a0, a1), (b0, b1) = span(x), span(conditional_input)
            if a0 < b1 and b0 < a1:
                raise PlanNotApplicable("x and conditional_input overlap in memory (column views of one tensor): a plan rebinds them separately")
        self.pdf, self.kwargs, self.dev = pdf, dict(kwargs), dev
        self.sig = self._signature(x, conditional_input)
        # the status words of the replays live in pinned host memory (the kernels write them there when they have something to say): they
        # accumulate over the replays and are examined lazily, without a copy-back on the stream
        self.host_status, self.status = _hip.mapped_status(dev)
        self.pool = torch.cuda.MemPool()
        self._record(x, conditional_input)

    @staticmethod
    def _signature(x, c):
        # (+ the 16-byte alignment of the buffers: the entry points choose vector-load kernel variants by it at record time)
        return (tuple(x.shape), x.stride(), x.dtype, x.device, x.data_ptr() % 16,
                None if c is None else (tuple(c.shape), c.stride(), c.dtype, c.device, c.data_ptr() % 16))

    def _param_key(self):
        # the in-place version counters of the pdf's parameters.  Walking nn.Module.parameters() costs ~25 us per call for the 26 tensors of C3 (half
        # of a shard step's host time, scripts/probe/gather_cost.py): the tensor list is kept.  A parameter OBJECT that is replaced (module.weight =
        # nn.Parameter(...), load_state_dict(assign=True), parametrisation swaps) goes through nn.Module.register_parameter, whose global hook bumps
        # _PARAM_REGISTRATIONS: the list is walked again on the next call (it used to be noticed only every 256th call).  The periodic
        # walk stays as a back-stop for edits that bypass register_parameter (module._parameters[...] = ...); pdf.invalidate_packed_caches() is the
        # explicit way.
        n = self._key_calls = getattr(self, "_key_calls", 0) + 1
        ps = getattr(self, "_key_params", None)
        reg = _PARAM_REGISTRATIONS[0]
        if ps is None or reg != getattr(self, "_key_reg", -1) or (n & 255) == 0:
            self._key_reg = reg
            fresh = list(self.pdf.parameters())
            if ps is None or len(fresh) != len(ps) or any(a is not b for a, b in zip(fresh, ps)):
                self._key_gen = getattr(self, "_key_gen", 0) + 1
                self._key_params = ps = fresh
        return (self._key_gen,) + tuple([p._version for p in ps])

    def _record(self, x, cond):
        pdf = self.pdf
        dbg = os.environ.get("JF_PLAN_DEBUG")

        def say(*a):
            if dbg:
                torch.cuda.synchronize()
                print("[plan]", *a, flush=True)
        with torch.no_grad():
            saved = pdf.use_step_plans
            pdf.use_step_plans = False
            try:
                for _ in range(2):      # every lazily built cache (permanent rows, packed images, kernel attributes) exists before the recording
                    ref = pdf.forward(x, conditional_input=cond, **self.kwargs)
                pdf.flush_status()
                say("warm-up done")
                self.key = self._param_key()
                plan = _hip.StepPlan()
                pdf._capture_status = self.status
                try:
                    # intermediate buffers of the recorded pass must keep their addresses for the life of the plan: a private memory pool
                    # (no garbage collection while allocations are routed to it: see _RETIRED_POOLS)
                    import gc
                    gc.collect()
                    release_plan_memory()
                    gc_was_on = gc.isenabled()
                    gc.disable()
                    with torch.cuda.use_mem_pool(self.pool, device=self.dev), _RecordingPass() as rec:
                        plan.begin()
                        try:
                            # everything between begin() and end() aborts the recording on failure: a plan left recording keeps this
                            # thread's launches going into it instead of to the GPU
                            out = pdf.forward(x, conditional_input=cond, **self.kwargs)
                            if rec.foreign:
                                raise PlanNotApplicable("the step runs torch operations between the library's launches: %s" % sorted(set(rec.foreign)))
                            try:
                                self.slot_x = plan.add_slot(x)
                                self.slot_c = plan.add_slot(cond) if cond is not None else None
                                self.slot_out = [plan.add_slot(t) for t in out]
                            except RuntimeError as e:     # overlapping slots (x and cond views of one tensor), empty tensors
                                raise PlanNotApplicable("the inputs / outputs cannot be declared as plan slots: %s" % e)
                            plan.end()
                        except BaseException:
                            plan.abort()
                            raise
                finally:
                    pdf._capture_status = None
                    if "gc_was_on" in locals() and gc_was_on:
                        gc.enable()
                self.plan, self.out_like = plan, out
                if dbg:
                    import ctypes
                    buf = (ctypes.c_uint64 * 512)()
                    for op in range(plan.n_ops):
                        n = int(_hip.lib().jf_plan_debug_words(plan.handle, op, buf, 512))
                        print("[plan] op", op, " ".join("%x" % buf[i] for i in range(n) if buf[i] >> 40), flush=True)
                    print("[plan] pool tensors alive:", [(hex(t.data_ptr()), t.numel() * t.element_size()) for t in out], "status", hex(self.status.data_ptr()),
                          "host", hex(self.host_status.data_ptr()), flush=True)
                say("recorded", plan.n_ops, "ops", int(_hip.lib().jf_plan_num_relocations(plan.handle)), "relocations", plan.calls,
                    [hex(t.data_ptr()) for t in out], hex(x.data_ptr()))
                # self-check on DIFFERENT inputs (rows rotated by one): a step that left the library between two launches (a torch op on
                # the way) would replay that part with the recorded pass's values
                x2 = torch.roll(x, 1, 0)
                c2 = None if cond is None else torch.roll(cond, 1, 0)
                if x2.stride() != x.stride():
                    x2 = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device).copy_(x2)
                if c2 is not None and c2.stride() != cond.stride():
                    c2 = torch.empty_strided(cond.shape, cond.stride(), dtype=cond.dtype, device=cond.device).copy_(c2)
                want = pdf.forward(x2, conditional_input=c2, **self.kwargs)
                pdf.flush_status()
                say("eager on rotated rows done")
                got = self._replay(x2, c2)
                torch.cuda.synchronize(self.dev)
                say("replay on rotated rows done")
                for g, w in zip(got, want):
                    if not bool(((g == w) | (g.isnan() & w.isnan())).all()):
                        raise PlanNotApplicable("replaying the recorded step does not reproduce pdf.forward for this configuration")
                torch.cuda.synchronize(self.dev)
                self.host_status.zero_()
            finally:
                pdf.use_step_plans = saved

    def _replay(self, x, cond, stream=None, logp_out=None):
        out = [torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device) for t in self.out_like]
        if logp_out is not None:                        # the caller's buffer takes the place of the log-prob output (e.g. a slot of an exchange stage)
            t = self.out_like[0]
            if (tuple(logp_out.shape), logp_out.stride(), logp_out.dtype, logp_out.device) != (tuple(t.shape), t.stride(), t.dtype, t.device):
                raise ValueError("logp_out must be a %s tensor of shape %s, strides %s on %s" % (t.dtype, tuple(t.shape), t.stride(), t.device))
            out[0] = logp_out
        tensors = [x] + ([cond] if self.slot_c is not None else []) + out
        self._last_stream = stream if stream is not None else torch.cuda.current_stream(self.dev)
        self.plan.launch(tensors, self.dev, self._last_stream)
        return tuple(out)

    def __call__(self, x, conditional_input=None, stream=None, logp_out=None):
        """stream (optional): the CURRENT stream, when the caller already holds it (PipelinedForward: saves the look-ups); logp_out (optional): the
        tensor the step writes its log-probs into instead of a fresh one"""
        if self._signature(x, conditional_input) != self.sig:
            raise ValueError("this plan was recorded for inputs %s, got %s" % (self.sig, self._signature(x, conditional_input)))
        pdf = self.pdf
        if pdf.check_status and self.host_status.any():                 # words a finished replay copied back: non-zero = a problem in some earlier step
            self.flush()
        if self._param_key() != self.key:                               # parameters updated in place since the recording: the caches moved
            self._record(x, conditional_input)
        out = self._replay(x, conditional_input, stream, logp_out)
        if pdf.check_status and pdf.check_status != "deferred":
            self.flush()
        return out

    def flush(self):
        """wait for the replays so far and raise / warn as the eager path does (the status words accumulate over the replays)"""
        (getattr(self, "_last_stream", None) or torch.cuda.current_stream(self.dev)).synchronize()
        if self.host_status.any():
            words = self.host_status.clone()
            self.host_status.zero_()                    # (the stream is idle: no kernel is adding to the words)
            self.pdf._report_status(words)


_PIPELINE_STREAMS = {}


def _pipeline_streams(dev, depth):
    pool = _PIPELINE_STREAMS.setdefault((dev.type, dev.index), [])
    while len(pool) < depth:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:depth]


class PendingStep:
    """one submitted step of a PipelinedForward: `outputs` = (log_prob, log_prob_base, base), being computed on `stream`; `event` fires when they
    are complete.  result() makes the CALLER's current stream wait for them (not the host) and returns them."""
    __slots__ = ("outputs", "event", "stream")

    def __init__(self, outputs, event, stream):
        self.outputs, self.event, self.stream = outputs, event, stream

    def result(self):
        cur = torch.cuda.current_stream(self.outputs[0].device)
        if cur != self.stream:
            cur.wait_event(self.event)
            for t in self.outputs:
                t.record_stream(cur)                    # (allocated in the step's stream pool, used on the caller's stream)
        return self.outputs


class PipelinedForward:
    """pdf.forward for a stream of INDEPENDENT batches of one input signature (see pdf.pipelined_forward): consecutive steps alternate between
    `depth` HIP streams, each through its own recorded step plan (own intermediate buffers, own status words).  The last round of workgroups
    of one step's fused block leaves most of the chip idle (1.33 rounds at 2^17 rows: a third of the step); with the next step already queued
    on another stream, its first launches fill that tail -- the step plan of 2^17 rows 0.108 -> 0.089 ms, of 2^20 rows 0.70 -> 0.66 ms per
    step, results bit-identical (scripts/probe/two_stream.py).  The reference evaluates batches one call after the other
    (main/default.py:1059-1117); nothing couples them."""

    def __init__(self, pdf, x, conditional_input, depth, kwargs):
        if depth < 1:
            raise ValueError("depth >= 1")
        self.dev = _hip.require_device(x, conditional_input if isinstance(conditional_input, torch.Tensor) else None)
        self.pdf, self.depth, self.i = pdf, depth, 0
        caller = torch.cuda.current_stream(self.dev)
        # the step streams are shared by every PipelinedForward of the process (per device): the HIP runtime maps streams onto a few hardware
        # queues (4 by default, GPU_MAX_HW_QUEUES), and streams that share a queue serialise -- a process that made a fresh set of streams for every
        # pipeline (bench.py's rows sweep: one per batch size) measured 0.099 ms for the 2^17-row step that takes 0.089 in a fresh process
        self.streams = _pipeline_streams(self.dev, depth) if depth > 1 else [caller]
        self.plans = []
        # the two side blocks in ONE launch (merge_max_rows) pay on one stream at every size (2^20 rows: 0.714 vs 0.730 ms) and, with steps on
        # alternating streams, up to 2^19 rows (2^17: 0.106 vs 0.115); at 2^20 rows two separate launches interleave better with the neighbour
        # step's fused block (0.662 vs 0.667-0.69 ms, same-box A/B): recorded accordingly
        keep = pdf.merge_max_rows
        if depth > 1:
            pdf.merge_max_rows = min(keep, pdf.merge_max_rows_pipelined)
        try:
            for s in self.streams:
                s.wait_stream(caller)
                with torch.cuda.stream(s):
                    self.plans.append(PlannedForward(pdf, x, conditional_input, kwargs))
                caller.wait_stream(s)
        finally:
            pdf.merge_max_rows = keep

    def peek_stream(self):
        """the stream the NEXT submit() will run on"""
        return self.streams[self.i % self.depth]

    def submit(self, x, conditional_input=None, logp_out=None):
        """enqueue one step; returns a PendingStep at once.  The step starts when the work queued so far on the caller's current stream (the
        producer of x) is done; the caller's stream does NOT wait for the step -- PendingStep.result() / drain() do that.  logp_out: a buffer
        for the step's log-probs (parallel.PipelinedGather.next_slot(): the step writes straight into the stage of the next exchange)."""
        j = self.i % self.depth
        self.i += 1
        s = self.streams[j]
        cur = torch.cuda.current_stream(self.dev)
        if s != cur:
            s.wait_stream(cur)
        # (torch.cuda.set_stream there and back: the `with torch.cuda.stream(s)` context costs ~10 us of a shard step's ~55 us of host time)
        torch.cuda.set_stream(s)
        try:
            out = self.plans[j](x, conditional_input, stream=s, logp_out=logp_out)
            ev = torch.cuda.Event()
            ev.record(s)
        finally:
            torch.cuda.set_stream(cur)
        if s != cur:
            x.record_stream(s)
            if conditional_input is not None:
                conditional_input.record_stream(s)
        return PendingStep(out, ev, s)

    def drain(self):
        """the caller's current stream waits for every submitted step; then the plans' status words are examined (raises / warns as eager)"""
        cur = torch.cuda.current_stream(self.dev)
        for s in self.streams:
            if s != cur:
                cur.wait_stream(s)
        for p in self.plans:
            if self.pdf.check_status:
                p.flush()


class GraphedForward:
    """pdf.forward captured in a HIP graph for one input shape (see pdf.graphed_forward)."""

    def __init__(self, pdf, x, conditional_input, kwargs):
        _hip.require_device(x, conditional_input if isinstance(conditional_input, torch.Tensor) else None)
        self.pdf = pdf
        self.x = x.detach().clone()
        self.cond = None if conditional_input is None else conditional_input.detach().clone()
        self.kwargs = dict(kwargs)
        self.status = _hip.new_status(x.device)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(2):          # every lazily built cache (permanent rows, packed images, kernel attributes) exists before the capture
                pdf.forward(self.x, conditional_input=self.cond, **self.kwargs)
        torch.cuda.current_stream(x.device).wait_stream(side)
        pdf.flush_status()
        self.graph = torch.cuda.CUDAGraph()
        pdf._capture_status = self.status
        try:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.out = pdf.forward(self.x, conditional_input=self.cond, **self.kwargs)
        finally:
            pdf._capture_status = None

    def __call__(self, x, conditional_input=None, check=True):
        if x.shape != self.x.shape or x.dtype != self.x.dtype:
            raise ValueError("graphed_forward was captured for inputs %s %s, got %s %s" % (tuple(self.x.shape), self.x.dtype, tuple(x.shape), x.dtype))
        self.x.copy_(x)
        if self.cond is not None:
            self.cond.copy_(conditional_input)
        self.graph.replay()
        if check and self.pdf.check_status:
            self.pdf._report_status(self.status)         # one host read-back, raises / warns as the eager path does
        return self.out
