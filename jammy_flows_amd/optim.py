"""Adam for the models of this package in one launch per step (csrc/misc_kernels.hip: jf_adam_step).

The reference trains with ``torch.optim.Adam(pdf.parameters())`` (examples/jammy_flows.py:381-412, docs/source/usage/training.rst:24-44).  These
models have < 1 MB of parameters in ~40 small tensors; torch's foreach implementation spends 6-7 ``multi_tensor_apply`` launches of 10-20 us on
them -- 0.09 ms of a 1.85 ms C3 training step.  ``Adam`` here is a ``torch.optim.Optimizer`` with the same defaults, state names
(``step``, ``exp_avg``, ``exp_avg_sq``: torch state_dicts load) and update, issued as ONE launch per dtype.  Not supported (use torch's):
``amsgrad``, ``weight_decay``, ``maximize``, sparse gradients, parameters that are not on a HIP device."""
import torch

from . import _hip


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_dtype = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or not p.is_cuda or p.dtype not in (torch.float32, torch.float64) or not p.is_contiguous():
                    raise RuntimeError("jammy_flows_amd.optim.Adam: dense contiguous float32 / float64 parameters on a HIP device only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                by_dtype.setdefault((p.dtype, p.device, st["step"]), []).append((p, p.grad.contiguous(), st))
            for (dtype, dev, step), items in by_dtype.items():
                for i in range(0, len(items), _hip.JF_ADAM_MAX_TENSORS):
                    chunk = items[i:i + _hip.JF_ADAM_MAX_TENSORS]
                    arr = (_hip.jf_adam_tensor * len(chunk))()
                    for j, (p, g, st) in enumerate(chunk):
                        arr[j] = _hip.jf_adam_tensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                    _hip._launch("jf_adam_step" + _hip._suffix(chunk[0][0]), "n%d" % len(chunk),
                                 (arr, len(chunk), float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]), step), dev)
        # the kernel wrote the parameters behind torch's back: move their version counters, which the packed-weight caches (main/default.py) and
        # autograd's saved-tensor checks follow, as an in-place torch op would have
        touched = [p for group in self.param_groups for p in group["params"] if p.grad is not None]
        if touched:
            torch._C._autograd._unsafe_set_version_counter(touched, [p._version + 1 for p in touched])
        return loss
