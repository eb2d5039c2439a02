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


class AmortizableMLPSpec:
    """U/V/bias bookkeeping of AmortizableMLP._initialize_uv_structure (amortizable_mlp.py:272-375)."""

    def __init__(self, in_dim, hidden, out_dim, ranks):
        hidden = list_from_str(hidden) if isinstance(hidden, str) else ([hidden] if isinstance(hidden, int) else list(hidden))
        n_mat = len(hidden) + 1
        if isinstance(ranks, int):
            ranks = [ranks] * n_mat
        elif isinstance(ranks, str):
            ranks = list_from_str(ranks)
        assert len(ranks) == n_mat
        self.ins = [in_dim] + hidden
        self.outs = hidden + [out_dim]
        self.stages = []
        n = 0
        for i, (a, b) in enumerate(zip(self.ins, self.outs)):
            max_rank = min(a, b)
            used = min(max_rank, ranks[i]) if ranks[i] > 0 else max_rank
            full = not ((used * (a + b) < a * b) and ranks[i] > 0)
            nu = a * b if full else used * b
            nv = 0 if full else used * a
            self.stages.append(dict(inp=a, out=b, rank=used, full=full, nu=nu, nv=nv, nb=b))
            n += nu + nv + b
        self.num_amortization_params = n

    def apply(self, x, uvb):
        """_apply_amortized_mlp (amortizable_mlp.py:508-578); uvb (1|B, n)."""
        c = 0
        for i, st in enumerate(self.stages):
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
            x = x + b
            if i < len(self.stages) - 1:
                x = np.tanh(x)
        return x


class AmortizableMLP:
    def __init__(self, spec, uvb=None):
        self.spec = spec
        self.uvb = None if uvb is None else np.asarray(uvb, dtype=np.float64).reshape(1, -1)
        self.num_amortization_params = spec.num_amortization_params

    def __call__(self, x, extra_inputs=None):
        return self.spec.apply(x, self.uvb if extra_inputs is None else extra_inputs)
