"""AmortizableMLP -- an MLP whose U / V / bias entries live in ONE flat vector (``u_v_b_pars``), optionally with low-rank weight
matrices W = U V^T.  Same constructor, attribute names and flat-vector layout as jammy_flows/amortizable_mlp.py
(:11-260 construction, :272-375 layout, :377-484 initialisation, :508-682 forward), so a reference ``state_dict`` loads.

Compute: every matrix product runs on the matrix cores through ``jf_linear_*`` (csrc/mlp_kernels.hip); low-rank stages are two
launches (V^T first, then U with bias + tanh fused).  Only ``highway_mode=0`` (a plain MLP, the mode jammy_flows.pdf uses by
default) has a kernel path.
"""
import math

import numpy
import torch
from torch import nn

from . import _hip, autograd
from .extra_functions import list_from_str


class AmortizableMLP(nn.Module):
    def __init__(self, input_dim, hidden_dims, output_dim, highway_mode=0, low_rank_approximations=0, nonlinearity="tanh",
                 use_permanent_parameters=True, svd_mode="smart", precise_mlp_structure=dict()):
        super().__init__()
        if highway_mode != 0:
            raise NotImplementedError("AmortizableMLP highway_mode %d has no HIP path (only mode 0)" % highway_mode)
        if nonlinearity != "tanh":
            raise NotImplementedError("AmortizableMLP nonlinearity %s has no HIP path (only tanh)" % nonlinearity)
        if svd_mode not in ("smart", "naive"):
            raise Exception("unknown svd mode", svd_mode)
        if len(precise_mlp_structure) > 0:
            raise NotImplementedError("precise_mlp_structure is not supported")
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.highway_mode = highway_mode
        self.use_permanent_parameters = use_permanent_parameters
        self.nonlinearity = nonlinearity
        self.svd_mode = svd_mode
        if type(hidden_dims) == str:
            self.hidden_dims = list_from_str(hidden_dims)
        elif type(hidden_dims) == int:
            self.hidden_dims = [hidden_dims]
        elif type(hidden_dims) == list:
            self.hidden_dims = list(hidden_dims)
        else:
            raise Exception("Unsupported type ", type(hidden_dims), " for hidden_dims .. can be int/str/list of ints")
        n_mat = len(self.hidden_dims) + 1
        if type(low_rank_approximations) == int:
            ranks = [low_rank_approximations] * n_mat
        elif type(low_rank_approximations) == str:
            ranks = list_from_str(low_rank_approximations)
        else:
            ranks = list(low_rank_approximations)
        assert len(ranks) == n_mat
        self.total_low_rank_approximations = ranks

        # one entry per dense stage: flat-vector layout [U | V | b] per stage (amortizable_mlp.py:284-375)
        ins = [input_dim] + self.hidden_dims
        outs = self.hidden_dims + [output_dim]
        self.stages = []
        n = 0
        for i, (a, b) in enumerate(zip(ins, outs)):
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
            self.stages.append(dict(inp=a, out=b, rank=used, full=full, num_u=nu, num_v=nv, num_b=b, offset=n))
            n += nu + nv + b
        self.num_amortization_params = n
        self._cast_cache = None
        if use_permanent_parameters:
            self.u_v_b_pars = nn.Parameter(torch.randn(self.num_amortization_params).type(torch.double).unsqueeze(0))
            self.initialize_uvbs()

    # ---- initialisation (amortizable_mlp.py:377-484)
    def obtain_default_init_tensor(self, fix_final_bias=None, prev_damping_factor=1000.0):
        init = torch.randn(self.num_amortization_params, dtype=torch.float64).unsqueeze(0)
        for st in self.stages:
            if st["full"]:
                fan_in = st["inp"]
                gain = nn.init.calculate_gain("leaky_relu", numpy.sqrt(5))
                bound = math.sqrt(3.0) * gain / math.sqrt(fan_in)
                o = st["offset"]
                with torch.no_grad():
                    init[:, o:o + st["num_u"]].uniform_(-bound, bound)
                    bb = 1 / numpy.sqrt(fan_in)
                    init[:, o + st["num_u"]:o + st["num_u"] + st["num_b"]].uniform_(-bb, bb)
        if fix_final_bias is not None:
            init = init / prev_damping_factor
            init[0, -self.stages[-1]["num_b"]:] = fix_final_bias
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

    def forward(self, i, extra_inputs=None):
        """i: (B, input_dim).  extra_inputs (per-sample U/V/b vectors, amortize_everything) has no kernel yet."""
        if extra_inputs is not None:
            raise NotImplementedError("AmortizableMLP with per-sample weights (extra_inputs) has no HIP kernel yet")
        assert self.use_permanent_parameters
        _hip.require_device(i)
        grad = autograd._needs_grad(i, self.u_v_b_pars)
        flat = self.u_v_b_pars.to(dtype=i.dtype).reshape(-1) if grad else self._flat(i)
        lin = autograd.linear if grad else _hip.linear
        x = i
        last = len(self.stages) - 1
        for si, st in enumerate(self.stages):
            o = st["offset"]
            u = flat[o:o + st["num_u"]]
            v = flat[o + st["num_u"]:o + st["num_u"] + st["num_v"]]
            b = flat[o + st["num_u"] + st["num_v"]:o + st["num_u"] + st["num_v"] + st["num_b"]]
            act = 0 if si == last else 1
            if st["full"]:
                x = lin(x, u.view(st["out"], st["inp"]), b, act)
            else:
                t = lin(x, v.view(st["rank"], st["inp"]), None, 0)                # V^T x
                x = lin(t, u.view(st["out"], st["rank"]), b, act)                 # U (V^T x) + b
        return x
