#!/usr/bin/env python3
"""can the kernels' status words live in pinned host memory (device atomics over the bus, no copy-back launch)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch, fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
torch.set_grad_enabled(False)
host = torch.zeros(_hip.JF_STATUS_WORDS, dtype=torch.int32).pin_memory()


class Holder:
    def __init__(self, t):
        self.t = t
        self.__cuda_array_interface__ = {"shape": tuple(t.shape), "typestr": "<i4", "data": (t.data_ptr(), False), "version": 2}


dev_view = torch.as_tensor(Holder(host), device="cuda")
print("view", dev_view.device, dev_view.data_ptr() == host.data_ptr(), dev_view)
fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
x64, c64 = inputs(fx, 1 << 12, 3)
x = torch.from_numpy(x64).to("cuda", torch.float32)
x[5, 0] = float("nan")
x[77, 5] = float("inf")
pdf._capture_status = dev_view
try:
    out = pdf.forward(x)
finally:
    pdf._capture_status = None
torch.cuda.synchronize()
print("host words after a step with 2 bad rows:", host.tolist())
ref = _hip.new_status(x.device)
pdf._capture_status = ref
try:
    out = pdf.forward(x)
finally:
    pdf._capture_status = None
torch.cuda.synchronize()
print("device words:", ref.tolist())
