"""torch.autograd wrappers around the C-ABI launches: what makes ``-pdf(x)[0].mean().backward()`` work (the reference's training step,
examples/jammy_flows.py:381-412, docs/source/usage/training.rst:24-44).

Every Function's forward is the SAME kernel launch the inference path uses; nothing but the inputs is saved.  Backward:
  * g-layer chains: one hand-written HIP launch (jf_gf_chain_inv_bwd_*, csrc/gf_bwd_kernels.hip) that re-runs the chain and returns the
    gradient of the targets and of the parameter row block -- (B, P) for per-sample blocks, partial sums for permanent parameters;
  * dense layers: g W (row-parallel) runs on split-bf16 MFMA for aligned float32 shapes (jf_linear_split), on the MFMA dense kernel with the
    transposed weight in float64, and in the library only for what is left; the two products that reduce over the BATCH (g^T x, column sums)
    run in jf_linear_wgrad / jf_linear_wgrad_split (csrc/wgrad_kernels.hip, csrc/split_gemm_kernels.hip: the batch split over the grid -- the
    library's output-tiled GEMM walks 1e5..1e6 rows in a handful of workgroups there, 14 ms per call in float64); the tanh derivative is one
    launch on the saved activation (jf_tanh_bwd);
  * the fused conditional block (MLP + g layers in one launch): its gradient-mode forward keeps every layer's input coordinate and mixture
    sums (20 floats per row and layer), its backward is ONE launch (jf_cond_gf_chain_inv_split_bwd: parameters recomputed in MFMA registers,
    each layer's adjoint in place, g_h accumulated from the same registers) + the weight-gradient product on the packed gradient rows + the
    hidden layer's one-launch backward; JF_FUSED_BLOCK_BACKWARD=0 selects the round-2 sequence (parameter block recomputed by two dense
    launches, then the same two steps as a per-sample block).
"""
import os

import torch

from . import _hip


def _side_stream_safe(backward):
    """backward of a node that may run on one of the pdf's training side streams (default.pdf.train_streams): its saved tensors and incoming
    gradients were (possibly) allocated on the caller's stream and are released on the host as soon as the node returns, while its kernels may
    still be reading them -- they are kept alive until the streams have met again (_hip.keep_if_side_stream; a no-op on the caller's stream)."""
    def wrapped(ctx, *grads):
        if _hip.SIDE_STREAMS:
            _hip.keep_if_side_stream(*ctx.saved_tensors, *grads)
        out = backward(ctx, *grads)
        if _hip.SIDE_STREAMS:
            _hip.keep_if_side_stream(*(out if isinstance(out, tuple) else (out,)))     # side-stream allocations read by nodes on the caller's stream
        return out
    wrapped.__doc__ = backward.__doc__
    return wrapped


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def _input_grad(g, weight):
    """g (B, N) @ weight (N, K): the row-parallel product of a dense layer's backward.  float32 with aligned shapes: split-bf16 MFMA
    (jf_linear_split); float64: the MFMA dense kernel on the transposed weight (the library picks 128 x 128 macro tiles for these
    (B x 1224) (1224 x 8) shapes: 0.15 ms per call at 2^17 rows, six calls per C5 training step); else the library."""
    wt = weight.t()
    if _hip.linear_split_ok(g, wt):
        return _hip.linear_split(g, wt)
    if g.dtype == torch.float64 and g.shape[0] > 0:
        return _hip.linear(g, wt.contiguous(), None, 0)
    return g @ weight


class LinearFn(torch.autograd.Function):
    """out = act(inp @ weight^T + bias) on the MFMA dense kernel (jf_linear); act 0 identity / 1 tanh."""

    @staticmethod
    def forward(ctx, inp, weight, bias, act):
        out = _hip.linear(inp.detach(), weight.detach(), None if bias is None else bias.detach(), act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(inp, weight, out if act else None)
        return out

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g):
        inp, weight, out = ctx.saved_tensors
        g = _hip.tanh_bwd(g, out) if ctx.act else g.contiguous()
        g_inp = _input_grad(g, weight) if ctx.needs_input_grad[0] else None
        g_w = g_b = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            g_w, g_b = _hip.linear_wgrad(g, inp, want_bias=ctx.has_bias and ctx.needs_input_grad[2])
        return g_inp, (g_w if ctx.needs_input_grad[1] else None), g_b, None


def linear(inp, weight, bias=None, act=0):
    if _needs_grad(inp, weight, bias):
        return LinearFn.apply(inp, weight, bias, act)
    return _hip.linear(inp, weight, bias, act)


class ActivationFn(torch.autograd.Function):
    """AmortizableMLP nonlinearity other than tanh on a layer's pre-activation (jf_activation / jf_activation_bwd)."""

    @staticmethod
    def forward(ctx, z, code):
        ctx.code = code
        ctx.save_for_backward(z)
        return _hip.activation(z.detach(), code)

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return _hip.activation_bwd(g, z, ctx.code), None


def activation(z, code):
    if torch.is_grad_enabled() and z.requires_grad:
        return ActivationFn.apply(z, code)
    return _hip.activation(z, code)


class Mlp2SmallFn(torch.autograd.Function):
    """narrow Linear - tanh - Linear head (<= 32 inputs, <= 16 outputs; <= 8 inputs, <= 64 outputs) whose input rows need no gradient: forward = the fused jf_mlp2 launch,
    backward = ONE launch (jf_mlp2_small_bwd) instead of tanh', two weight gradients, two bias sums and g W2 of the per-layer path."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        ctx.save_for_backward(x, w1, b1, w2)
        return _hip.mlp2(x.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach())

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g):
        x, w1, b1, w2 = ctx.saved_tensors
        g_w1, g_b1, g_w2, g_b2 = _hip.mlp2_small_bwd(x, w1, b1, w2, g)
        return None, g_w1, g_b1, g_w2, g_b2


def mlp2_small_ok(x, lin1, lin2):
    return (not x.requires_grad and _hip.mlp2_small_shape_ok(lin1.in_features, lin1.out_features, lin2.out_features, x.element_size())
            and lin1.out_features % 4 == 0 and lin1.bias is not None and lin2.bias is not None)


class CombineRowsFn(torch.autograd.Function):
    """(total, base_logp) = (sum of the blocks' log-dets + sum of their base log-probs, the latter) of a gradient-mode forward whose blocks ran on
    side streams (main/default.py: _forward_with_grad): ONE launch (jf_combine_rows) where torch ran one add per pair and one for the total --
    five launches of ~5 us each on the step's longest chain for three blocks; the backward hands the incoming gradients through (no launch
    unless both outputs carry one)."""

    @staticmethod
    def forward(ctx, n_ld, *parts):
        ld, blp = parts[:n_ld], parts[n_ld:]
        ctx.n = (n_ld, len(blp))
        ctx.set_materialize_grads(False)
        _, blp_sum, total = _hip.combine_rows([t.detach() for t in ld], [t.detach() for t in blp], want_total=True)
        return total, blp_sum

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_total, g_blp):
        n_ld, n_blp = ctx.n
        g_b = g_total if g_blp is None else (g_blp if g_total is None else g_total + g_blp)
        return (None,) + (g_total,) * n_ld + (g_b,) * n_blp


class GfChainInvFn(torch.autograd.Function):
    """log-prob direction of a chain of g layers (jf_gf_chain_inv) -> (x_out, log_det_out, base_logp_out)."""

    @staticmethod
    def forward(ctx, x, log_det, params, base_logp_in, layer_array, n_layers, D, status):
        res = _hip.gf_chain("inv", x.detach(), None if log_det is None else log_det.detach(), params.detach(), layer_array, n_layers, D,
                            base_logp_in=None if base_logp_in is None else base_logp_in.detach(), want_base_logp=True, status=status)
        ctx.meta = (layer_array, n_layers, D, status)
        ctx.set_materialize_grads(False)             # unused outputs arrive as None (the kernels take NULL), not as zero-filled tensors
        ctx.save_for_backward(x, params)
        ctx.has = (log_det is not None, base_logp_in is not None)
        return res

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_xout, g_ld, g_blp):
        x, params = ctx.saved_tensors
        layer_array, n_layers, D, status = ctx.meta
        g_x, g_params = _hip.gf_chain_inv_bwd(x, params, layer_array, n_layers, D, g_xout, g_ld, g_blp, status=None)
        return (g_x, g_ld if ctx.has[0] else None, g_params, g_blp if ctx.has[1] else None, None, None, None, None)


class InverseJacobianFn(torch.autograd.Function):
    """delta = J^-1 r for the Jacobian J = dy/dx of the log-prob direction at a SOLUTION of y(x) = z (r = y - z: zero to solver precision, so the
    forward returns zeros); what matters is the backward, lambda = J^-T g -- the adjoint of sampling by the implicit-function theorem
    (the reference differentiates through its Newton iterations instead, bisection_n_newton.py:74-93).  J is block lower triangular over the
    autoregressive sub-pdfs (block j sees the earlier blocks' coordinates through its conditioning MLP), J^T block upper triangular:

        for block j = last .. first:   lambda_j = J_jj^-T (g_j - sum_{k > j} (dy_k / dx_j)^T lambda_k)

    J_jj^-T v: one co-vector launch through the block's g layers (blocks[j]["cot"]), or -- manifold / 't' blocks of 1-3 dimensions -- the dense
    (B, d, d) block from d vector-Jacobian products; the coupling sum is the x-gradient of ONE input-gradient pass through block k with
    upstream lambda_k.  Autograd then continues into the graph of y with lambda: the parameter gradients need no further work here."""

    @staticmethod
    def forward(ctx, r, x0, blocks):
        ctx.x0, ctx.blocks = x0, blocks
        return torch.zeros_like(r)

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g):
        x0, blocks = ctx.x0, ctx.blocks
        rhs = g.clone()
        lam = torch.empty_like(g)
        with torch.enable_grad():
            for blk in reversed(blocks):
                a, b, y = blk["a"], blk["b"], blk["y"]
                v = rhs[:, a:b].contiguous()
                lj = blk["cot"](v) if blk["cot"] is not None else None
                if lj is None:
                    rows = [torch.autograd.grad(y[:, k].sum(), x0, retain_graph=True)[0][:, a:b] for k in range(b - a)]
                    J = torch.stack(rows, dim=1)                                       # J[n, k, i] = d y_k / d x_i inside the block
                    lj = torch.linalg.solve(J.transpose(1, 2), v.unsqueeze(-1)).squeeze(-1)
                lam[:, a:b] = lj
                if blk["coupled"] and a > 0:
                    (gx,) = torch.autograd.grad(y, x0, grad_outputs=lj, retain_graph=True)
                    rhs[:, :a] -= gx[:, :a]
        return lam, None, None


class _SplitFlatFn(torch.autograd.Function):
    """a flat parameter vector cut into consecutive pieces in ONE graph node.  Slicing the vector piece by piece (`flat[a:b]` per weight
    matrix, as the AmortizableMLP stages do) costs the backward pass a zero-filled copy of the whole vector per piece plus the additions
    that accumulate them: 30 launches of ~5 us per C5 training step for the two low-rank MLPs (rocprofv3, round 5).  Here the backward is one
    concatenation."""

    @staticmethod
    def forward(ctx, flat, sizes):
        ctx.sizes = sizes
        ctx.set_materialize_grads(False)
        return tuple(flat.detach().split(sizes))

    @staticmethod
    @_side_stream_safe
    def backward(ctx, *gs):
        like = next((g for g in gs if g is not None), None)
        if like is None:
            return None, None
        return torch.cat([g.reshape(-1) if g is not None else like.new_zeros(n) for g, n in zip(gs, ctx.sizes)]), None


class FlatPieces:
    """`flat[a:b]` for the (a, b) of a fixed list of cut points, served from one _SplitFlatFn node; any other slice falls back to slicing the
    vector itself (correct, just the slow pattern above)."""

    def __init__(self, flat, cuts):
        self.flat = flat
        sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
        self._pieces = _SplitFlatFn.apply(flat, sizes)
        self._index = {(a, b): k for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))}
        self.dtype, self.device = flat.dtype, flat.device

    def element_size(self):
        return self.flat.element_size()

    def __getitem__(self, sl):
        if isinstance(sl, slice) and sl.step is None:
            k = self._index.get((sl.start or 0, sl.stop))
            if k is not None:
                return self._pieces[k]
        return self.flat[sl]


class LowRankHeadFn(torch.autograd.Function):
    """t2 = V2 tanh(U1 (V1 c) + b1): the two-stage low-rank AmortizableMLP up to the input of its last U product (float64), forward and backward
    one launch each (csrc/jf_lowrank_mlp.h) instead of three dense launches forward and eight dense / elementwise launches backward."""

    @staticmethod
    def forward(ctx, inp, v1, u1, b1, v2):
        t2, t1, h = _hip.lowrank_head(inp.detach(), v1.detach(), u1.detach(), b1.detach(), v2.detach())
        ctx.save_for_backward(inp, v1, u1, v2, t1, h)
        return t2

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_t2):
        inp, v1, u1, v2, t1, h = ctx.saved_tensors
        g_inp, g_v1, g_u1, g_b1, g_v2 = _hip.lowrank_head_bwd(inp, v1, u1, v2, t1, h, g_t2, ctx.needs_input_grad[0])
        return g_inp, g_v1, g_u1, g_b1, g_v2


def lowrank_head(inp, v1, u1, b1, v2):
    return LowRankHeadFn.apply(inp, v1, u1, b1, v2)


class LowRankGfChainFn(torch.autograd.Function):
    """chain of g layers on the rows u2 t2 + b2 of a low-rank last MLP stage (float64, rank <= 8), the (B, N) parameter block and its gradient
    never materialised: forward jf_lowrank_gf_chain_inv (keeps each layer's inputs and mixture sums, 320 bytes per row and layer), backward
    jf_lowrank_gf_chain_inv_bwd (per-layer launches that contract the parameter gradients with u2 and t2 on the matrix cores)."""

    @staticmethod
    def forward(ctx, t2, u2, b2, x, log_det, base_logp_in, layer_array, n_layers, D, status):
        res = _hip.lowrank_gf_chain_inv(t2.detach(), u2.detach(), b2.detach(), x.detach(), None if log_det is None else log_det.detach(), layer_array,
                                        n_layers, D, base_logp_in=None if base_logp_in is None else base_logp_in.detach(), want_base_logp=True,
                                        want_aux=True, status=status)
        if res is None:
            raise RuntimeError("jf_lowrank_gf_chain_inv: unsupported configuration (the caller checks lowrank_chain_ok first)")
        ctx.meta = (layer_array, n_layers, D)
        ctx.set_materialize_grads(False)
        ctx.has = (log_det is not None, base_logp_in is not None)
        ctx.aux = res[3]
        ctx.save_for_backward(t2, u2, b2, res[0])
        return res[:3]

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_xout, g_ld, g_blp):
        t2, u2, b2, z = ctx.saved_tensors
        layer_array, n_layers, D = ctx.meta
        # (ctx.aux stays: differentiable sampling runs several backward passes through the same graph, main/default.py: _differentiable_sample)
        g_x, g_t2, g_u2, g_b2 = _hip.lowrank_gf_chain_inv_bwd(t2, u2, b2, ctx.aux, z, layer_array, n_layers, D, g_xout, g_ld, g_blp)
        return (g_t2, g_u2, g_b2, g_x, g_ld if ctx.has[0] else None, g_blp if ctx.has[1] else None, None, None, None, None)


# the fused block's backward in one launch (csrc/cond_bwd_kernels.hip) wherever the "split" kernel ran the forward; "0" = the round-2 sequence
# of dense launches around a materialised parameter block (kept for A/B timing and as the path of the other kernels)
FUSED_BLOCK_BACKWARD = os.environ.get("JF_FUSED_BLOCK_BACKWARD", "1") != "0"
# its weight-gradient product on f16 pairs (three MFMA passes) instead of bf16 triples (six): JF_WGRAD_F16_PAIRS=0 selects the triples
WGRAD_F16_PAIRS = os.environ.get("JF_WGRAD_F16_PAIRS", "1") != "0"
# the blocks' log-det / base log-prob sums and the total of a gradient-mode forward in one launch (CombineRowsFn): JF_COMBINE_ROWS_FN=0 -> torch adds
COMBINE_ROWS_FN = os.environ.get("JF_COMBINE_ROWS_FN", "1") != "0"
_packed_row_index = {}


def _packed_rows(layer_array, n_layers, D, device):
    """device index tensor: natural parameter column -> column of the packed gradient row of the fused backward kernel"""
    idx = _hip.cond_gf_packed_rows(layer_array, n_layers, D)
    key = (tuple(idx), str(device))
    t = _packed_row_index.get(key)
    if t is None:
        t = _packed_row_index[key] = torch.tensor(idx, dtype=torch.int64, device=device)
    return t


class CondBlockFn(torch.autograd.Function):
    """conditional e-block in one launch (jf_cond_gf_chain_inv[_split]): amortisation MLP Linear-tanh-Linear + its g layers."""

    @staticmethod
    def forward(ctx, inp, w1, b1, w2, b2, x, log_det, base_logp_in, packed, layer_array, n_layers, D, status):
        args = (x.detach(), None if log_det is None else log_det.detach(), layer_array, n_layers, D)
        kw = dict(base_logp_in=None if base_logp_in is None else base_logp_in.detach(), want_base_logp=True, status=status)
        aux = None
        if packed is not None:
            if FUSED_BLOCK_BACKWARD and packed[0] in ("split", "split16") and w1.shape[0] % 4 == 0 and x.shape[0] > 0:
                aux = _hip.cond_gf_aux(x.shape[0], n_layers, x.device)
            res = _hip.cond_gf_chain_inv_split(inp.detach(), w1.detach(), b1.detach(), packed[1], *args, kind=packed[0], aux=aux, **kw)
        else:
            res = _hip.cond_gf_chain_inv(inp.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach(), *args, **kw)
        ctx.meta = (layer_array, n_layers, D)
        ctx.set_materialize_grads(False)             # unused outputs arrive as None (the kernels take NULL), not as zero-filled tensors
        ctx.has = (log_det is not None, base_logp_in is not None)
        ctx.fused = None if aux is None else (packed, aux)
        ctx.save_for_backward(inp, w1, b1, w2, b2, x, res[0] if aux is not None else None)
        return res

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_xout, g_ld, g_blp):
        inp, w1, b1, w2, b2, x, z = ctx.saved_tensors
        layer_array, n_layers, D = ctx.meta
        need = ctx.needs_input_grad
        if ctx.fused is not None:
            # ONE launch: hidden activations and parameters recomputed in the forward kernel's register layout, each layer's adjoint in place,
            # g_h accumulated from the same registers; the parameter-row gradient leaves in packed column order for the weight-gradient product
            (kind, packed), aux = ctx.fused
            if kind != "split16":                     # the adjoint kernel multiplies f16 pairs: its own image of the same weights
                packed = _hip.cond_gf_pack(w2, b2, layer_array, n_layers, D, "split16")
            packed_t = _hip.cond_gf_bwd_pack(w2, layer_array, n_layers, D)
            f16_wgrad = WGRAD_F16_PAIRS and (need[3] or need[4]) and w1.shape[0] % 4 == 0 and 16 < w1.shape[0] <= 128 and inp.shape[0] >= 4096
            res = _hip.cond_gf_chain_inv_split_bwd(inp, w1, b1, packed, packed_t, z, aux, layer_array, n_layers, D, g_xout, g_ld, g_blp,
                                                   want_absmax=f16_wgrad)
            g_x, g_p, h, g_hid = res[:4]
            ctx.fused = None
            del aux
            g_w2 = g_b2 = None
            # (round 6: the hidden layer's adjoint forked onto a second stream beside the weight-gradient product -- independent launches,
            #  0.04 ms off this node's chain on paper -- made the graph-replayed C3 training step 0.10 ms SLOWER, 1.282 -> 1.385 ms: a third stream in
            #  the captured step costs more than the overlap returns; profiles/r06_experiments.md)
            if need[3] or need[4]:
                if f16_wgrad:                          # the packed rows' largest entry comes with them: weight gradient on f16 pairs; its slab
                    # sum puts the packed rows back in natural order (no gather launches)
                    g_w2, g_b2 = _hip.linear_wgrad_split16(g_p, h, res[4], 14, want_bias=need[4],
                                                           rows=tuple(_hip.cond_gf_packed_rows(layer_array, n_layers, D)))
                else:
                    rows = _packed_rows(layer_array, n_layers, D, g_p.device)
                    g_w2, g_b2 = _hip.linear_wgrad(g_p, h, want_bias=need[4])
                    g_w2 = g_w2.index_select(0, rows)
                    g_b2 = None if g_b2 is None else g_b2.index_select(0, rows)
        else:
            # the parameter block is not kept by the forward launch: two dense launches bring it back (the large one on split-bf16 MFMA, float32:
            # the arithmetic of the fused forward block)
            h = _hip.linear(inp, w1, b1, 1)
            split = _hip.linear_split_ok(h, w2, b2)
            params = _hip.linear_split(h, w2, b2) if split else _hip.linear(h, w2, b2, 0)
            g_x, g_p = _hip.gf_chain_inv_bwd(x, params, layer_array, n_layers, D, g_xout, g_ld, g_blp, status=None)
            del params
            g_w2, g_b2 = _hip.linear_wgrad(g_p, h, want_bias=need[4]) if (need[3] or need[4]) else (None, None)
            g_hid = None
        if not need[0] and inp.shape[1] <= _hip.MLP2_SMALL_MAX_IN and w1.shape[0] <= _hip.MLP2_MAX_HIDDEN and inp.shape[0] > 0:
            # data rows in front: tanh derivative + first-layer weight / bias gradient in one launch (the activations recomputed per hidden unit)
            g_w1, g_b1 = _hip.mlp_hidden_bwd(inp, w1, b1, g_hid if g_hid is not None else _input_grad(g_p, w2))
            g_inp = None
        else:
            g_h = _hip.tanh_bwd(g_hid if g_hid is not None else _input_grad(g_p, w2), h, inplace=True)
            g_w1, g_b1 = _hip.linear_wgrad(g_h, inp, want_bias=need[2]) if (need[1] or need[2]) else (None, None)
            g_inp = _input_grad(g_h, w1) if need[0] else None
        del g_p
        return (g_inp, g_w1, g_b1, g_w2, g_b2, g_x, g_ld if ctx.has[0] else None, g_blp if ctx.has[1] else None, None, None, None, None, None)


def embed(tgt, kind, layer):
    """differentiable conditioning embedding of a target block (main/default.py:946-962; sphere_base.py:305-332): only needed when the
    TARGET itself requires grad (d log_prob / d x through the autoregressive conditioning); plain elementwise torch ops."""
    if kind != "s" or tgt.shape[1] != layer.dimension:
        return tgt
    if layer.dimension == 1:
        return torch.cat([torch.cos(tgt), torch.sin(tgt)], dim=1)
    th = tgt[:, 0:1].clamp(1e-7, 3.14159265358979323846 - 1e-7)
    ph = tgt[:, 1:2]
    st = torch.sin(th)
    return torch.cat([st * torch.cos(ph), st * torch.sin(ph), torch.cos(th)], dim=1)


class SphereEmbeddingFn(torch.autograd.Function):
    """S1 / S2 chart change angles <-> embedding coordinates (sphere_base.py:242-335) with its log-det term: forward = the kernel of the
    inference path (jf_sphere_to/from_embedding), backward = the closed-form Jacobian applied to the incoming gradients.  What training in
    embedding coordinates (pdf(x_xyz, force_embedding_coordinates=True)) back-propagates through ahead of the block loop."""

    @staticmethod
    def forward(ctx, x, log_det, dim, to_embedding):
        out, ld = _hip.sphere_embedding(x.detach(), None if log_det is None else log_det.detach(), dim, to_embedding)
        ctx.meta = (dim, to_embedding, log_det is not None)
        ctx.save_for_backward(x.detach(), out)
        ctx.set_materialize_grads(False)
        if dim != 2 and log_det is None:
            ld = None
        return out, ld

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_out, g_ld):
        dim, to_embedding, has_ld = ctx.meta
        x, out = ctx.saved_tensors
        g_x = None
        if dim == 1:
            if to_embedding:                     # phi -> (cos phi, sin phi)
                if g_out is not None:
                    g_x = (-g_out[:, 0:1] * out[:, 1:2] + g_out[:, 1:2] * out[:, 0:1])
            elif g_out is not None:              # (x, y) -> phi: d phi = (x dy - y dx) / rho^2
                rho2 = (x * x).sum(dim=1, keepdim=True)
                g_x = g_out * torch.cat([-x[:, 1:2], x[:, 0:1]], dim=1) / rho2
        else:
            if to_embedding:                     # (theta, phi) -> (sin t cos p, sin t sin p, cos t), log_det += log sin t
                st, ct = torch.sin(x[:, 0:1]), torch.cos(x[:, 0:1])
                sp, cp = torch.sin(x[:, 1:2]), torch.cos(x[:, 1:2])
                g_t = torch.zeros_like(st)
                g_p = torch.zeros_like(st)
                if g_out is not None:
                    g_t = g_out[:, 0:1] * ct * cp + g_out[:, 1:2] * ct * sp - g_out[:, 2:3] * st
                    g_p = -g_out[:, 0:1] * st * sp + g_out[:, 1:2] * st * cp
                if g_ld is not None:
                    g_t = g_t + g_ld.unsqueeze(1) * ct / st
                g_x = torch.cat([g_t, g_p], dim=1)
            else:                                # (x, y, z) -> theta = acos(z / r), phi; log_det -= log sin theta
                rho2 = (x[:, 0:2] ** 2).sum(dim=1, keepdim=True)
                r2 = rho2 + x[:, 2:3] ** 2
                rho = rho2.sqrt()
                g_t = torch.zeros_like(rho) if g_out is None else g_out[:, 0:1]
                if g_ld is not None:
                    g_t = g_t - g_ld.unsqueeze(1) * x[:, 2:3] / rho          # d(-log sin theta)/d theta = -cot theta = -z / rho
                dt = torch.cat([x[:, 0:1] * x[:, 2:3] / (r2 * rho), x[:, 1:2] * x[:, 2:3] / (r2 * rho), -rho / r2], dim=1)
                g_x = g_t * dt
                if g_out is not None:
                    g_x = g_x + g_out[:, 1:2] * torch.cat([-x[:, 1:2] / rho2, x[:, 0:1] / rho2, torch.zeros_like(rho)], dim=1)
        return g_x, (g_ld if has_ld else None), None, None


def sphere_embedding(x, log_det, dim, to_embedding):
    """chart change with a graph when one is needed, the plain launch otherwise; log_det: tensor, None or a python number (= offset)"""
    ld_t = log_det if isinstance(log_det, torch.Tensor) else None
    if torch.is_grad_enabled() and _needs_grad(x, ld_t):
        out, ld = SphereEmbeddingFn.apply(x, ld_t, dim, to_embedding)
        if dim != 2:
            return out, log_det
        if ld_t is None and log_det is not None and log_det != 0:
            ld = ld + log_det
        return out, ld
    return _hip.sphere_embedding(x, log_det, dim, to_embedding)


class MChainInvFn(torch.autograd.Function):
    """log-prob direction of a chain of manifold layers (jf_{r,o,m,f,v,c}_chain_inv) -> (x_out, log_det_out, base_logp_out).
    Backward: jf_*_chain_inv_jvp launches (forward-mode passes through the very same device code instantiated on dual numbers)."""

    @staticmethod
    def forward(ctx, x, log_det, params, base_logp_in, fam, structs, dim, status):
        res = _hip.mchain(fam, "inv", x.detach(), None if log_det is None else log_det.detach(), params.detach(), structs, dim,
                          base_logp_in=None if base_logp_in is None else base_logp_in.detach(), want_base_logp=True, status=status)
        ctx.meta = (fam, structs, dim)
        ctx.set_materialize_grads(False)             # unused outputs arrive as None (the kernels take NULL), not as zero-filled tensors
        ctx.has = (log_det is not None, base_logp_in is not None)
        ctx.save_for_backward(x, params)
        return res

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_xout, g_ld, g_blp):
        x, params = ctx.saved_tensors
        fam, structs, dim = ctx.meta
        g_x, g_params = _hip.mchain_inv_bwd(fam, x, params, structs, dim, g_xout, g_ld, g_blp)
        return (g_x, g_ld if ctx.has[0] else None, g_params, g_blp if ctx.has[1] else None, None, None, None, None)


class TLayerInvFn(torch.autograd.Function):
    """log-prob direction of the affine 't' layer (jf_t_layer_inv) -> (x_out, log_det_out, base_logp_out)"""

    @staticmethod
    def forward(ctx, x, log_det, params, base_logp_in, struct, D, status):
        res = _hip.t_layer("inv", x.detach(), None if log_det is None else log_det.detach(), None if params is None else params.detach(), struct, D,
                           base_logp_in=None if base_logp_in is None else base_logp_in.detach(), want_base_logp=True, status=status)
        ctx.meta = (struct, D)
        ctx.set_materialize_grads(False)             # unused outputs arrive as None (the kernels take NULL), not as zero-filled tensors
        ctx.has = (log_det is not None, base_logp_in is not None, params is not None)
        ctx.save_for_backward(x, params)
        return res

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g_xout, g_ld, g_blp):
        x, params = ctx.saved_tensors
        struct, D = ctx.meta
        g_x, g_params = _hip.t_layer_inv_bwd(x, params, struct, D, g_xout, g_ld, g_blp)
        return (g_x, g_ld if ctx.has[0] else None, g_params if ctx.has[2] else None, g_blp if ctx.has[1] else None, None, None, None)


class AmlpStageFn(torch.autograd.Function):
    """one AmortizableMLP stage with per-sample weights (jf_amlp_stage): out = act(W_b x_b + bias_b) [+ residual]"""

    @staticmethod
    def forward(ctx, x, seg, residual, n_in, n_out, rank, has_bias, act):
        y = _hip.amlp_stage(x.detach(), seg.detach(), n_in, n_out, rank, has_bias, act, None)
        ctx.meta = (n_in, n_out, rank, has_bias, act)
        ctx.save_for_backward(x, seg, y if act else None)
        return y if residual is None else y + residual

    @staticmethod
    @_side_stream_safe
    def backward(ctx, g):
        x, seg, y = ctx.saved_tensors
        n_in, n_out, rank, has_bias, act = ctx.meta
        g_x, g_seg = _hip.amlp_stage_bwd(x, seg, n_in, n_out, rank, has_bias, act, y, g.contiguous(), want_g_in=ctx.needs_input_grad[0])
        return g_x, g_seg, (g if ctx.needs_input_grad[2] else None), None, None, None, None, None


def amlp_stage(x, seg, n_in, n_out, rank, has_bias, act, residual=None):
    if _needs_grad(x, seg, residual):
        return AmlpStageFn.apply(x, seg, residual, n_in, n_out, rank, has_bias, act)
    return _hip.amlp_stage(x, seg, n_in, n_out, rank, has_bias, act, residual)
