"""Adam for the models of this package in one launch per step (csrc/misc_kernels.hip: jf_adam_step).

The reference trains with ``torch.optim.Adam(pdf.parameters())`` (examples/jammy_flows.py:381-412, docs/source/usage/training.rst:24-44).  These
models have < 1 MB of parameters in ~40 small tensors; torch's foreach implementation spends 6-7 ``multi_tensor_apply`` launches of 10-20 us on
them -- 0.09 ms of a 1.85 ms C3 training step.  ``Adam`` here is a ``torch.optim.Optimizer`` with the same defaults, state names
(``step``, ``exp_avg``, ``exp_avg_sq``: torch state_dicts load) and update, issued as ONE launch per dtype.  Not supported (use torch's):
``amsgrad``, ``weight_decay``, ``maximize``, sparse gradients, parameters that are not on a HIP device.

``capturable=True`` (torch's name for it) keeps the step count in device memory -- one int64 tensor shared by the parameters that started
together, ``state[p]["step"]`` -- so that a training step captured in a HIP graph (``torch.cuda.graph``) replays with the right bias
corrections: ``step()`` is then one in-place add on that tensor and one ``jf_adam_step_dev`` launch, nothing on the host depends on the count."""
import torch

from . import _hip


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, capturable=bool(capturable)))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_dtype, new_counters = {}, {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or not p.is_cuda or p.dtype not in (torch.float32, torch.float64) or not p.is_contiguous():
                    raise RuntimeError("jammy_flows_amd.optim.Adam: dense contiguous float32 / float64 parameters on a HIP device only")
                st = self.state[p]
                capturable = group.get("capturable", False)
                if len(st) == 0:
                    if capturable:                                     # one device counter for the parameters that start in this call
                        if p.device not in new_counters:
                            new_counters[p.device] = torch.zeros((), dtype=torch.int64, device=p.device)
                        st["step"] = new_counters[p.device]
                    else:
                        st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if capturable:
                    if not torch.is_tensor(st["step"]) or st["step"].device != p.device or st["step"].dtype != torch.int64:
                        # a loaded state: a python int (non-capturable run) or one float32 tensor per parameter (what load_state_dict makes of
                        # "step" for a capturable group).  Reads the value on the host -- call step() once before capturing a graph.
                        val = int(st["step"])
                        if (p.device, val) not in new_counters:
                            new_counters[(p.device, val)] = torch.full((), val, dtype=torch.int64, device=p.device)
                        st["step"] = new_counters[(p.device, val)]
                    by_dtype.setdefault((p.dtype, p.device, st["step"].data_ptr()), []).append((p, p.grad.contiguous(), st))
                else:
                    st["step"] = int(st["step"]) + 1
                    by_dtype.setdefault((p.dtype, p.device, st["step"]), []).append((p, p.grad.contiguous(), st))
            bumped = set()
            for key, items in by_dtype.items():
                dtype, dev, step = key
                if group.get("capturable", False):
                    counter = items[0][2]["step"]
                    if counter.data_ptr() not in bumped:               # shared by the float32 and float64 parameters that started together
                        counter.add_(1)
                        bumped.add(counter.data_ptr())
                for i in range(0, len(items), _hip.JF_ADAM_MAX_TENSORS):
                    chunk = items[i:i + _hip.JF_ADAM_MAX_TENSORS]
                    arr = (_hip.jf_adam_tensor * len(chunk))()
                    for j, (p, g, st) in enumerate(chunk):
                        arr[j] = _hip.jf_adam_tensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                    hyper = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
                    if group.get("capturable", False):
                        _hip._launch("jf_adam_step_dev" + _hip._suffix(chunk[0][0]), "n%d" % len(chunk), (arr, len(chunk)) + hyper + (_hip._ptr(counter),), dev)
                    else:
                        _hip._launch("jf_adam_step" + _hip._suffix(chunk[0][0]), "n%d" % len(chunk), (arr, len(chunk)) + hyper + (step,), dev)
        # the kernel wrote the parameters behind torch's back: move their version counters, which the packed-weight caches (main/default.py) and
        # autograd's saved-tensor checks follow, as an in-place torch op would have
        touched = [p for group in self.param_groups for p in group["params"] if p.grad is not None]
        if touched:
            torch._C._autograd._unsafe_set_version_counter(touched, [p._version + 1 for p in touched])
        return loss
