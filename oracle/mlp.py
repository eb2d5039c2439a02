"""Parameter-emitting MLPs of the reference restated in numpy (oracle = test infrastructure only):
the default nn.Sequential(Linear, tanh, ..., Linear) (main/default.py:656-670) and AmortizableMLP
(jammy_flows/amortizable_mlp.py, highway_mode 0, svd_mode "smart")."""
import numpy as np


def list_from_str(spec):
    """extra_functions.py:91-95."""
    if spec == "":
        return []
    return [int(s) for s in spec.split("-")]


class SequentialMLP:
    def __init__(self, sd, prefix, n_linear):
        self.layers = [(np.asarray(sd["%s%d.weight" % (prefix, 2 * i)], dtype=np.float64),
                        np.asarray(sd["%s%d.bias" % (prefix, 2 * i)], dtype=np.float64)) for i in range(n_linear)]

    def __call__(self, x, extra_inputs=None):
        for i, (w, b) in enumerate(self.layers):
            x = x @ w.T + b
            if i < len(self.layers) - 1:
                x = np.tanh(x)
        return x


def _stage_layout(inputs, outputs, ranks, add_final_bias):
    """U / V / bias sizes of one sub-MLP (_initialize_uv_structure, amortizable_mlp.py:272-375, svd_mode "smart")"""
    stages, n = [], 0
    for i, (a, b) in enumerate(zip(inputs, outputs)):
        max_rank = min(a, b)
        used = min(max_rank, ranks[i]) if ranks[i] > 0 else max_rank
        full = not ((used * (a + b) < a * b) and ranks[i] > 0)
        nu = a * b if full else used * b
        nv = 0 if full else used * a
        last = i == len(inputs) - 1
        nb = b if (not last or add_final_bias) else 0
        stages.append(dict(inp=a, out=b, rank=used, full=full, nu=nu, nv=nv, nb=nb, act=not last))
        n += nu + nv + nb
    return stages, n


class AmortizableMLPSpec:
    """structure of an AmortizableMLP (amortizable_mlp.py:44-260): highway_mode 0 = plain MLP; 1 = MLP + linear skip connection (which carries
    the final bias); 2 / 3 / 4 = a sum of one-hidden-layer MLPs whose inputs are the MLP input / the running output / both, + the linear part.
    The linear part's parameters are the LAST ones of the flat vector (:621-629)."""

    def __init__(self, in_dim, hidden, out_dim, ranks, highway_mode=0):
        hidden = list_from_str(hidden) if isinstance(hidden, str) else ([hidden] if isinstance(hidden, int) else list(hidden))
        nh = len(hidden)
        n_mat = nh + 1 if highway_mode == 0 else (nh + 2 if highway_mode == 1 else 2 * nh + 1)
        if isinstance(ranks, int):
            ranks = [ranks] * n_mat
        elif isinstance(ranks, str):
            ranks = list_from_str(ranks)
        assert len(ranks) == n_mat
        self.highway_mode = highway_mode
        self.subs, self.linear = [], None
        if highway_mode == 0:
            self.subs.append(_stage_layout([in_dim] + hidden, hidden + [out_dim], ranks, True) + ("in",))
        elif highway_mode == 1:
            if nh > 0:
                self.subs.append(_stage_layout([in_dim] + hidden, hidden + [out_dim], ranks[:-1], False) + ("in",))
            self.linear = _stage_layout([in_dim], [out_dim], ranks[-1:], True)
        else:
            start = {2: in_dim, 3: out_dim, 4: in_dim + out_dim}[highway_mode]
            kind = {2: "in", 3: "out", 4: "in+out"}[highway_mode]
            for ind in range(nh):
                a = in_dim if ind == 0 else start
                self.subs.append(_stage_layout([a, hidden[ind]], [hidden[ind], out_dim], ranks[2 * ind:2 * ind + 2], False)
                                 + ("in" if ind == 0 else kind,))
            self.linear = _stage_layout([in_dim], [out_dim], ranks[-1:], True)
        self.num_amortization_params = sum(n for _, n, _ in self.subs) + (self.linear[1] if self.linear else 0)
        self.stages = self.subs[0][0] if highway_mode == 0 else None

    @staticmethod
    def _run(stages, x, uvb, c):
        """_apply_amortized_mlp (amortizable_mlp.py:508-578); uvb (1|B, n)."""
        for st in stages:
            u = uvb[:, c:c + st["nu"]]; c += st["nu"]
            v = uvb[:, c:c + st["nv"]]; c += st["nv"]
            b = uvb[:, c:c + st["nb"]]; c += st["nb"]
            if st["full"]:
                a = u.reshape(-1, st["out"], st["inp"])
                x = np.einsum("bij,bj->bi", np.broadcast_to(a, (x.shape[0],) + a.shape[1:]), x)
            else:
                um = u.reshape(u.shape[0], st["out"], st["rank"])
                vm = v.reshape(v.shape[0], st["rank"], st["inp"])
                t = np.einsum("bij,bj->bi", np.broadcast_to(vm, (x.shape[0],) + vm.shape[1:]), x)
                x = np.einsum("bij,bj->bi", np.broadcast_to(um, (x.shape[0],) + um.shape[1:]), t)
            if st["nb"]:
                x = x + b
            if st["act"]:
                x = np.tanh(x)
        return x, c

    def apply(self, x, uvb):
        """forward (amortizable_mlp.py:581-682)"""
        prev = 0.0
        if self.linear is not None:
            prev, _ = self._run(self.linear[0], x, uvb, self.num_amortization_params - self.linear[1])
        c = 0
        for stages, n, kind in self.subs:
            inp = x if kind == "in" else (prev if kind == "out" else np.concatenate([x, prev], axis=1))
            out, c = self._run(stages, inp, uvb, c)
            prev = prev + out
        return prev


class AmortizableMLP:
    def __init__(self, spec, uvb=None):
        self.spec = spec
        self.uvb = None if uvb is None else np.asarray(uvb, dtype=np.float64).reshape(1, -1)
        self.num_amortization_params = spec.num_amortization_params

    def __call__(self, x, extra_inputs=None):
        return self.spec.apply(x, self.uvb if extra_inputs is None else extra_inputs)
