"""AmortizableMLP -- an MLP whose U / V / bias entries live in ONE flat vector (``u_v_b_pars``), optionally with low-rank weight
matrices W = U V^T and with the reference's five "highway" layouts.  Same constructor, attribute names and flat-vector layout as
jammy_flows/amortizable_mlp.py (:11-260 construction, :272-375 layout, :377-484 initialisation, :508-682 forward), so a reference
``state_dict`` loads, and the same call convention: permanent parameters, or -- ``use_permanent_parameters=False`` -- per-sample
parameters handed in as ``extra_inputs`` (B, num_amortization_params) by a hyper-network (``fully_amortized_pdf``).

Compute: permanent weights -> every matrix product on the matrix cores through ``jf_linear_*`` (csrc/mlp_kernels.hip; low-rank stages are
two launches, V^T first, then U with bias + tanh fused).  Per-sample weights -> ``jf_amlp_stage_*`` (csrc/amlp_kernels.hip), a streaming pass
over the parameter block (no weight is shared between rows, so there is nothing to put on the matrix cores).  Both are wrapped in autograd
Functions (jammy_flows_amd/autograd.py) when gradients are needed."""
import math

import numpy
import torch
import os

from torch import nn

from . import _hip, autograd
from .extra_functions import NONLINEARITIES, list_from_str


def _stage_layout(inputs, outputs, ranks, add_final_bias, svd_mode):
    """per-stage bookkeeping of one sub-MLP (amortizable_mlp.py:272-375): [U | V | b] sizes, full-matrix flags, total"""
    stages, total = [], 0
    for i, (a, b) in enumerate(zip(inputs, outputs)):
        max_rank = min(a, b)
        if ranks[i] > 0:
            used = min(max_rank, ranks[i])
        else:
            used = 0 if svd_mode == "naive" else max_rank
        if svd_mode == "naive":
            full = used == 0
        else:
            full = not ((used * (a + b) < a * b) and ranks[i] > 0)
        nu = a * b if full else used * b
        nv = 0 if full else used * a
        last = i == len(inputs) - 1
        nb = b if (not last or add_final_bias) else 0
        stages.append(dict(inp=a, out=b, rank=0 if full else used, full=full, num_u=nu, num_v=nv, num_b=nb, act=0 if last else 1))
        total += nu + nv + nb
    return stages, total


class AmortizableMLP(nn.Module):
    def __init__(self, input_dim, hidden_dims, output_dim, highway_mode=0, low_rank_approximations=0, nonlinearity="tanh",
                 use_permanent_parameters=True, svd_mode="smart", precise_mlp_structure=dict()):
        super().__init__()
        if nonlinearity not in NONLINEARITIES:
            raise KeyError(nonlinearity)
        if svd_mode not in ("smart", "naive"):
            raise Exception("unknown svd mode", svd_mode)
        assert 0 <= highway_mode <= 4
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.highway_mode = highway_mode
        self.use_permanent_parameters = use_permanent_parameters
        self.nonlinearity = nonlinearity
        self.svd_mode = svd_mode
        # gradient mode, float64, two low-rank stages: the head in one launch forward / backward (JF_LOWRANK_HEAD=0: the per-stage launches)
        self.head_one_launch = os.environ.get("JF_LOWRANK_HEAD", "1") != "0"
        if len(precise_mlp_structure) > 0:
            # the caller hands in the sub-MLP table itself (amortizable_mlp.py:20, 56-62): per-matrix ranks, widths no `hidden_dims` string gives
            self._init_from_table(precise_mlp_structure)
            return
        if type(hidden_dims) == str:
            self.hidden_dims = list_from_str(hidden_dims)
        elif type(hidden_dims) == int:
            self.hidden_dims = [hidden_dims]
        elif type(hidden_dims) == list:
            self.hidden_dims = list(hidden_dims)
        else:
            raise Exception("Unsupported type ", type(hidden_dims), " for hidden_dims .. can be int/str/list of ints")
        nh = len(self.hidden_dims)
        n_mat = nh + 1 if highway_mode == 0 else (nh + 2 if highway_mode == 1 else 2 * nh + 1)
        if type(low_rank_approximations) == int:
            ranks = [low_rank_approximations] * n_mat
        elif type(low_rank_approximations) == str:
            ranks = list_from_str(low_rank_approximations)
        else:
            ranks = list(low_rank_approximations)
        assert len(ranks) == n_mat
        self.total_low_rank_approximations = ranks

        # sub-MLPs in flat-vector order; the linear highway (if any) sits at the very END of the vector (:621-629)
        self.sub_mlps = []          # list of (stages, n_params, input_kind) with input_kind in {"in", "out", "in+out"}
        self.linear = None
        if highway_mode < 2:
            if highway_mode == 0:
                st, n = _stage_layout([input_dim] + self.hidden_dims, self.hidden_dims + [output_dim], ranks, True, svd_mode)
                self.sub_mlps.append((st, n, "in"))
            else:
                if nh > 0:
                    st, n = _stage_layout([input_dim] + self.hidden_dims, self.hidden_dims + [output_dim], ranks[:-1], False, svd_mode)
                    self.sub_mlps.append((st, n, "in"))
                self.linear = _stage_layout([input_dim], [output_dim], ranks[-1:], True, svd_mode)
        else:
            start = {2: input_dim, 3: output_dim, 4: input_dim + output_dim}[highway_mode]
            kind = {2: "in", 3: "out", 4: "in+out"}[highway_mode]
            for ind in range(nh):
                a = input_dim if ind == 0 else start
                st, n = _stage_layout([a, self.hidden_dims[ind]], [self.hidden_dims[ind], output_dim], ranks[2 * ind:2 * ind + 2], False, svd_mode)
                self.sub_mlps.append((st, n, "in" if ind == 0 else kind))
            self.linear = _stage_layout([input_dim], [output_dim], ranks[-1:], True, svd_mode)
        self.num_amortization_params = sum(n for _, n, _ in self.sub_mlps) + (self.linear[1] if self.linear is not None else 0)
        self.stages = self.sub_mlps[0][0] if (highway_mode == 0) else None        # plain-MLP view used by the fused low-rank block
        self._cast_cache = None
        self._cut_points = None
        if use_permanent_parameters:
            self.u_v_b_pars = nn.Parameter(torch.randn(self.num_amortization_params).type(torch.double).unsqueeze(0))
            self.initialize_uvbs()

    def _init_from_table(self, table):
        """`precise_mlp_structure`: {"mlp_list": [sub-MLP dicts], "linear_highway": dict (highway modes > 0)}, each dict with `inputs`, `outputs`,
        `low_rank_approximations` (one entry per matrix), `add_final_bias`, `svd_mode` -- the reference's own `sub_mlp_structures` (:118-246).
        The reference keeps the activations as callables inside the table; the kernels know the named ones, so a table's `activations` (if
        present) must be the module's nonlinearity everywhere but after a sub-MLP's last matrix, as the reference itself fills them (:258-270)."""
        assert "mlp_list" in table
        if self.highway_mode > 0:
            assert "linear_highway" in table
        probe = torch.linspace(-2.0, 2.0, 9, dtype=torch.float64)
        named = {"tanh": torch.tanh, "relu": torch.relu, "softplus": torch.nn.functional.softplus, "elu": torch.nn.functional.elu,
                 "swish": lambda t: t * torch.sigmoid(t), "square": lambda t: t * t, "identity": lambda t: t}[self.nonlinearity]

        def check_acts(d):
            acts = d.get("activations")
            if not acts:
                return
            assert len(acts) == len(d["inputs"])
            for i, f in enumerate(acts):
                want = probe if i == len(acts) - 1 else named(probe)
                if not torch.allclose(f(probe), want, rtol=1e-12, atol=1e-12):
                    raise NotImplementedError("precise_mlp_structure: activation %d of a sub-MLP is neither the module's nonlinearity (%s) nor the "
                                              "identity after the last matrix" % (i, self.nonlinearity))
        self.hidden_dims = []
        self.sub_mlps, self.linear = [], None
        later = {0: "in", 1: "in", 2: "in", 3: "out", 4: "in+out"}[self.highway_mode]
        ranks_all = []
        for ind, d in enumerate(table["mlp_list"]):
            check_acts(d)
            ranks = [int(r) for r in d["low_rank_approximations"]]
            assert len(d["inputs"]) == len(d["outputs"]) == len(ranks)
            st, n = _stage_layout([int(v) for v in d["inputs"]], [int(v) for v in d["outputs"]], ranks, bool(d["add_final_bias"]), d.get("svd_mode", self.svd_mode))
            want_in = self.input_dim if (ind == 0 or later == "in") else (self.output_dim if later == "out" else self.input_dim + self.output_dim)
            assert st[0]["inp"] == want_in and st[-1]["out"] == self.output_dim, "sub-MLP %d does not fit highway_mode %d" % (ind, self.highway_mode)
            self.sub_mlps.append((st, n, "in" if ind == 0 else later))
            ranks_all += ranks
        if "linear_highway" in table:
            d = table["linear_highway"]
            check_acts(d)
            ranks = [int(r) for r in d["low_rank_approximations"]]
            self.linear = _stage_layout([int(v) for v in d["inputs"]], [int(v) for v in d["outputs"]], ranks, bool(d["add_final_bias"]), d.get("svd_mode", self.svd_mode))
            assert self.linear[0][0]["inp"] == self.input_dim and self.linear[0][-1]["out"] == self.output_dim
            ranks_all += ranks
        if self.highway_mode < 2:
            assert len(self.sub_mlps) <= 1
        self.total_low_rank_approximations = ranks_all
        self.num_amortization_params = sum(n for _, n, _ in self.sub_mlps) + (self.linear[1] if self.linear is not None else 0)
        self.stages = self.sub_mlps[0][0] if (self.highway_mode == 0 and self.sub_mlps) else None
        self._cast_cache = None
        self._cut_points = None
        if self.use_permanent_parameters:
            self.u_v_b_pars = nn.Parameter(torch.randn(self.num_amortization_params).type(torch.double).unsqueeze(0))
            self.initialize_uvbs()

    # ---- initialisation (amortizable_mlp.py:377-484)
    def obtain_default_init_tensor(self, fix_final_bias=None, prev_damping_factor=1000.0):
        init = torch.randn(self.num_amortization_params, dtype=torch.float64).unsqueeze(0)
        o = 0
        all_stages = [st for stages, _, _ in self.sub_mlps for st in stages]
        if self.linear is not None:
            all_stages += self.linear[0]
        for st in all_stages:
            if st["full"]:
                fan_in = st["inp"]
                gain = nn.init.calculate_gain("leaky_relu", numpy.sqrt(5))
                bound = math.sqrt(3.0) * gain / math.sqrt(fan_in)
                with torch.no_grad():
                    init[:, o:o + st["num_u"]].uniform_(-bound, bound)
                    if st["num_b"] > 0:
                        bb = 1 / numpy.sqrt(fan_in)
                        init[:, o + st["num_u"]:o + st["num_u"] + st["num_b"]].uniform_(-bb, bb)
            o += st["num_u"] + st["num_v"] + st["num_b"]
        if fix_final_bias is not None:
            init = init / prev_damping_factor
            nb = all_stages[-1]["num_b"]
            init[0, -nb:] = fix_final_bias
        return init.squeeze(0)

    def initialize_uvbs(self, fix_total=None, fix_final_bias=None, prev_damping_factor=1000.0):
        assert self.use_permanent_parameters, "Initialization of uvb tensor only makes sense for permanent parameters."
        if fix_total is not None:
            self.u_v_b_pars.data[0, ...] = fix_total
        else:
            self.u_v_b_pars.data[0, ...] = self.obtain_default_init_tensor(fix_final_bias=fix_final_bias, prev_damping_factor=prev_damping_factor)

    # ---- forward
    def _flat(self, like):
        p = self.u_v_b_pars
        key = (like.dtype, like.device, p._version, p.data_ptr())
        if self._cast_cache is None or self._cast_cache[0] != key:
            with torch.no_grad():
                self._cast_cache = (key, p.detach().to(device=like.device, dtype=like.dtype).reshape(-1).contiguous())
        return self._cast_cache[1]

    def _cuts(self):
        """boundaries of every u / v / b block of the parameter vector (the slices the stages take), sorted"""
        if self._cut_points is None:
            cuts = {0, self.num_amortization_params}

            def add(stages, o):
                for st in stages:
                    cuts.update((o, o + st["num_u"], o + st["num_u"] + st["num_v"], o + st["num_u"] + st["num_v"] + st["num_b"]))
                    o += st["num_u"] + st["num_v"] + st["num_b"]
            if self.linear is not None:
                add(self.linear[0], self.num_amortization_params - self.linear[1])
            o = 0
            for stages, n, kind in self.sub_mlps:
                add(stages, o)
                o += n
            self._cut_points = sorted(c for c in cuts if 0 <= c <= self.num_amortization_params)
        return self._cut_points

    def _grad_flat(self, like):
        """the permanent parameter vector inside the autograd graph, pre-cut at every block boundary (autograd.FlatPieces)"""
        flat = self.u_v_b_pars.to(dtype=like.dtype).reshape(-1)
        if flat.requires_grad and torch.is_grad_enabled():
            return autograd.FlatPieces(flat, self._cuts())
        return flat

    def lowrank_views(self, flat):
        """(v1, u1, b1, v2, u2, b2) views into the flat vector when this is a plain two-stage MLP whose last stage is low-rank and whose sizes
        fit jf_amlp2 / jf_amlp_gf_chain_inv (K1 <= 32, hidden <= 128, ranks <= 16); else None"""
        if self.highway_mode != 0 or self.stages is None or len(self.stages) != 2 or self.nonlinearity != "tanh":
            return None
        s1, s2 = self.stages
        if s2["full"] or s2["rank"] > 16 or (not s1["full"] and s1["rank"] > 16) or s1["inp"] > 32 or s1["out"] > 128:
            return None
        elem = flat.element_size()
        lds = ((s1["rank"] * s1["inp"] + s1["out"] * s1["rank"] if not s1["full"] else s1["out"] * s1["inp"]) + s1["out"] + s2["rank"] * s2["inp"]
               + s2["out"] * (s2["rank"] | 1) + s2["out"]) * elem
        if lds > 160 * 1024:
            return None
        o = 0
        if s1["full"]:
            v1, u1 = None, flat[o:o + s1["num_u"]].view(s1["out"], s1["inp"])
        else:
            u1 = flat[o:o + s1["num_u"]].view(s1["out"], s1["rank"])
            v1 = flat[o + s1["num_u"]:o + s1["num_u"] + s1["num_v"]].view(s1["rank"], s1["inp"])
        b1 = flat[o + s1["num_u"] + s1["num_v"]:o + s1["num_u"] + s1["num_v"] + s1["num_b"]]
        o += s1["num_u"] + s1["num_v"] + s1["num_b"]
        u2 = flat[o:o + s2["num_u"]].view(s2["out"], s2["rank"])
        v2 = flat[o + s2["num_u"]:o + s2["num_u"] + s2["num_v"]].view(s2["rank"], s2["inp"])
        b2 = flat[o + s2["num_u"] + s2["num_v"]:o + s2["num_u"] + s2["num_v"] + s2["num_b"]]
        return v1, u1, b1, v2, u2, b2

    def _run(self, stages, x, flat, o, per_sample, residual=None):
        """one sub-MLP.  flat: 1-d permanent vector or (B, P) per-sample block; o: offset of the sub-MLP's first parameter.
        `residual` is added to the result of the LAST stage (fused into the launch for per-sample weights)."""
        last = len(stages) - 1
        for si, st in enumerate(stages):
            res = residual if si == last else None
            n = st["num_u"] + st["num_v"] + st["num_b"]
            act = st["act"] if self.nonlinearity == "tanh" else 0          # tanh: fused into the launch; others: a pass on the pre-activation
            if per_sample:
                x = autograd.amlp_stage(x, flat[:, o:o + n], st["inp"], st["out"], st["rank"], st["num_b"] > 0, act, res)
            else:
                lin = autograd.linear
                u = flat[o:o + st["num_u"]]
                v = flat[o + st["num_u"]:o + st["num_u"] + st["num_v"]]
                b = flat[o + st["num_u"] + st["num_v"]:o + n] if st["num_b"] > 0 else None
                if st["full"]:
                    x = lin(x, u.view(st["out"], st["inp"]), b, act)
                else:
                    t = lin(x, v.view(st["rank"], st["inp"]), None, 0)                # V^T x
                    x = lin(t, u.view(st["out"], st["rank"]), b, act)                 # U (V^T x) + b
                if res is not None:
                    x = x + res
            if st["act"] and self.nonlinearity != "tanh":
                x = autograd.activation(x, _hip.ACT_CODES[self.nonlinearity])
            o += n
        return x, o

    def forward_to_last_rank(self, i):
        """(t, u, b) with t (B, rank) = V^T x of the LAST stage -- the rank-space vector of every row -- and that stage's u (out, rank), b (out)
        as views of the parameter vector, all three inside the autograd graph: for a caller that fuses u t + b into its own launch
        (autograd.LowRankGfChainFn, which never writes the (B, out) block).  None unless this is a plain MLP (no highway connections) on permanent
        parameters whose last stage is low-rank with a bias."""
        if not self.use_permanent_parameters or self.highway_mode != 0 or self.stages is None or self.linear is not None or len(self.sub_mlps) != 1:
            return None
        last = self.stages[-1]
        if last["full"] or last["num_b"] == 0 or last["act"]:
            return None
        _hip.require_device(i)
        flat = self._grad_flat(i)
        views = self.lowrank_views(flat) if self.head_one_launch else None
        if views is not None and _hip.lowrank_head_ok(i, *views[:4]):          # the whole head in one launch (csrc/jf_lowrank_mlp.h)
            v1, u1, b1, v2, u2, b2 = views
            return autograd.lowrank_head(i, v1, u1, b1, v2), u2, b2
        x, o = self._run(self.stages[:-1], i, flat, 0, False)
        u = flat[o:o + last["num_u"]].view(last["out"], last["rank"])
        v = flat[o + last["num_u"]:o + last["num_u"] + last["num_v"]].view(last["rank"], last["inp"])
        b = flat[o + last["num_u"] + last["num_v"]:o + last["num_u"] + last["num_v"] + last["num_b"]]
        return autograd.linear(x, v, None, 0), u, b

    def forward(self, i, extra_inputs=None):
        """i: (B, input_dim).  extra_inputs: None (permanent parameters) or the per-sample (B, num_amortization_params) block."""
        _hip.require_device(i)
        per_sample = extra_inputs is not None
        if per_sample:
            assert not self.use_permanent_parameters, "MLP uses permanent parameters but extra inputs are given in forward. This is not allowed!"
            assert extra_inputs.shape[1] == self.num_amortization_params, (
                "Extra inputs dimension (%d) does not match number of amortization params of MLP (%d) " % (extra_inputs.shape[1],
                                                                                                            self.num_amortization_params))
            flat = extra_inputs
        else:
            assert self.use_permanent_parameters
            grad = autograd._needs_grad(i, self.u_v_b_pars)
            flat = self._grad_flat(i) if grad else self._flat(i)
            views = self.lowrank_views(flat) if (not grad or self.head_one_launch) else None
            if views is not None and not grad:             # hidden-128 / rank-r MLP of the reference's custom mode: ONE launch (jf_amlp2)
                return _hip.amlp2(i, *views)
            if views is not None and _hip.lowrank_head_ok(i, *views[:4]):
                # gradient mode: the head (everything in front of the last U product) forward and backward in one launch each, then the last product
                v1, u1, b1, v2, u2, b2 = views
                return autograd.linear(autograd.lowrank_head(i, v1, u1, b1, v2), u2, b2, 0)
        prev = None
        if self.linear is not None:                            # its parameters are the LAST ones of the vector (:621-629)
            prev, _ = self._run(self.linear[0], i, flat, self.num_amortization_params - self.linear[1], per_sample)
        o = 0
        for mi, (stages, n, kind) in enumerate(self.sub_mlps):
            if kind == "in":
                inp = i
            elif kind == "out":
                inp = prev
            else:
                inp = torch.cat([i, prev], dim=1)
            prev, o = self._run(stages, inp, flat, o, per_sample, residual=prev)
        return prev
