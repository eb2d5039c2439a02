"""where do the sporadic ~10-50 ms stalls of the float32 fixture scan come from?  python3 scripts/probe/stall_probe.py [fixture] [f32|f64] [calls]
Per call: host time of the call; every 8 calls a synchronize, timed.  Outliers are printed with what the garbage collector did meanwhile."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import fixture_io, helpers

name = sys.argv[1] if len(sys.argv) > 1 else "r_i1"
DT = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
B = 1 << 16
GC_LOG = []


def on_gc(phase, info):
    if phase == "start":
        GC_LOG.append([info["generation"], time.perf_counter(), None])
    else:
        GC_LOG[-1][2] = time.perf_counter()


gc.callbacks.append(on_gc)
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, DT, torch.device("cuda"))
pdf.check_status = "deferred"
reps = (B + fx["x"].shape[0] - 1) // fx["x"].shape[0]
x = torch.from_numpy(np.tile(fx["x"][:-8], (reps + 1, 1))[:B]).to(device="cuda", dtype=DT)
c = None if fx.get("cond") is None else torch.from_numpy(np.tile(fx["cond"][:-8], (reps + 1, 1))[:B]).to(device="cuda", dtype=DT)
emb = bool(fx.meta.get("embedding"))


def fwd():
    with torch.no_grad():
        pdf(x, conditional_input=c, force_embedding_coordinates=emb)


def train():
    for p in pdf.parameters():
        p.grad = None
    with torch.enable_grad():
        (-pdf(x, conditional_input=c, force_embedding_coordinates=emb)[0].mean()).backward()


for label, fn, gc_on in (("forward", fwd, True), ("forward, gc disabled", fwd, False), ("train", train, True), ("train, gc disabled", train, False)):
    gc.enable() if gc_on else gc.disable()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    del GC_LOG[:]
    host, syncs = [], []
    t_all = time.perf_counter()
    for i in range(N):
        t0 = time.perf_counter()
        fn()
        host.append(time.perf_counter() - t0)
        if i % 8 == 7:
            t0 = time.perf_counter()
            torch.cuda.synchronize()
            syncs.append(time.perf_counter() - t0)
    t_all = time.perf_counter() - t_all
    host, syncs = np.array(host) * 1e3, np.array(syncs) * 1e3
    slow_h, slow_s = np.nonzero(host > 2.0)[0], np.nonzero(syncs > 2.0)[0]
    gcs = [(g, (b - a) * 1e3) for g, a, b in GC_LOG if b is not None]
    print("%s %s %s: %d calls in %.1f ms; host per call median %.3f max %.3f ms; sync median %.3f max %.3f ms" % (
        name, "f32" if DT == torch.float32 else "f64", label, N, t_all * 1e3, np.median(host), host.max(), np.median(syncs), syncs.max()))
    print("   calls above 2 ms: %s" % [(int(i), round(float(host[i]), 2)) for i in slow_h[:20]])
    print("   syncs above 2 ms: %s" % [(int(i) * 8 + 7, round(float(syncs[i]), 2)) for i in slow_s[:20]])
    print("   collections: %d (gen2: %d), longest %.2f ms" % (len(gcs), sum(1 for g, _ in gcs if g == 2), max([d for _, d in gcs] + [0.0])), flush=True)
