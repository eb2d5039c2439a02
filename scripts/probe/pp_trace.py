#!/usr/bin/env python3
"""Step timeline of the ping-pong fused block kernel (private -DPP_TRACE build of the library: scripts/probe/pp_trace.sh): s_memtime at every
step boundary of workgroup 0 -> per wave and step: cycles of work (start of step -> arrival at the barrier) and of waiting at the barrier."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import bench
import fixture_io
import helpers
from jammy_flows_amd import _hip

torch.set_grad_enabled(False)
W = bench.WORKLOADS["c3"]
B = 1 << 20
fx = fixture_io.load(W["fixture"])
pdf = helpers.build_product(fx, torch.float32, "cuda")
pdf.check_status = False
pdf.fused_block_kernel = "pp"
x64, _ = bench.make_inputs("c3", B, W["seed"])
x = torch.from_numpy(x64).to("cuda", torch.float32)
for _ in range(3):
    pdf(x)
torch.cuda.synchronize()
lib = _hip.lib()
n = 8 * 512
buf = (ctypes.c_longlong * n)()
lib.jf_pp_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.jf_pp_trace_read(buf, n) == 0
t = np.array(buf, dtype=np.int64).reshape(8, 512)
t0 = t[:, 0].min()
for w in (0, 4):
    st = t[w, 496:504]
    print('wave %d first-layer stamps (entry, state set, inputs split, MFMAs issued, tanh tile 0..3 done):' % w, [int(v - st[0]) for v in st[:8]])
t[:, 496:] = 0
names = ["Ma", "Mb", "Fa", "Fb"]
for w in (0, 4):
    tw = t[w]
    k = int((tw > 0).sum()) // 2
    print("wave %d: %d steps, total %d cycles" % (w, k, tw[2 * k - 1] - tw[0]))
    work = tw[0:2 * k:2][1:] - tw[1:2 * k:2][:-1]        # barrier release -> next arrival
    wait = tw[1:2 * k:2] - tw[0:2 * k:2]                 # arrival -> release
    off = 2 if w >= 4 else 0
    rows = []
    for i in range(min(k - 1, 40)):
        s = i + 1 - off
        rows.append("%s%d work %5d wait %5d" % (names[s % 4] if s >= 0 else "--", (s // 4) % 4 if s >= 0 else 0, work[i], wait[i + 1]))
    print("\n".join(rows))
    s_idx = np.arange(1, k) - off
    for kind in range(4):
        sel = (s_idx >= 0) & (s_idx % 4 == kind) & (np.arange(1, k) < k - 3)
        print("  %s: work mean %.0f (min %d max %d), wait after it mean %.0f" % (names[kind], work[sel].mean(), work[sel].min(), work[sel].max(),
                                                                                  wait[1:][sel].mean()))
