import sys, os
R = "/root/repo"; sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); sys.path.insert(0, R + "/tests/golden")
import numpy as np, torch
from test_gpu_parity import ALL_FIXTURES, build_product, to_dev
torch.set_grad_enabled(False)
from jammy_flows_amd import _hip
_hip.LIB_PATH = os.environ['JF_LIB']
fx = [f for f in ALL_FIXTURES if f.name == "c3_e4s2e4"][0]
n = fx["x"].shape[0]
reps = (1 << 18) // n + 2
big = reps * n - 41
pdf = build_product(fx, torch.float32); pdf.check_status = False
x = to_dev(np.tile(fx["x"], (reps, 1))[:big], torch.float32)
emb = bool(fx.meta["embedding"])
np.set_printoptions(precision=5, suppress=True, linewidth=220)
for it in range(int(os.environ.get("JF_ITERS", "6"))):
    r = pdf(x, force_embedding_coordinates=emb)
    logp, dbg = r[0].cpu().numpy(), r[1].cpu().numpy(); xd = r[2].cpu().numpy()[:, -4:]
    rs = pdf(x[:n], force_embedding_coordinates=emb)
    lps, dbgs = rs[0].cpu().numpy(), rs[1].cpu().numpy(); xds = rs[2].cpu().numpy()[:, -4:]
    ref, dref = np.tile(lps, reps)[:big], np.tile(dbgs, reps)[:big]
    fin = np.isfinite(logp) & np.isfinite(ref)
    bad = np.nonzero(fin & (np.abs(logp - ref) > 1e-3 * (1 + np.abs(ref))))[0]
    dbad = np.nonzero(np.isfinite(dbg) & np.isfinite(dref) & (np.abs(dbg - dref) > 1e-3 * (1 + np.abs(dref))))[0]
    print("iter", it, "dbg", os.environ.get("JF_CS_DBG"), "bad logp rows", len(bad), "bad dbg rows", len(dbad), "overlap", len(set(bad) & set(dbad)))
    xref = np.tile(xds, (reps, 1))[:big]
    xbad = np.nonzero((np.abs(xd - xref) > 1e-3 * (1 + np.abs(xref))).any(axis=1))[0]
    print('   per-lane dbg rows differing', len(xbad), 'overlap with bad logp', len(set(xbad) & set(bad)))
    if len(xbad):
        b = xbad[:5]; print('   lane dbg', xd[b], 'ref', xref[b])
    if len(bad):
        b = bad[:6]
        print("   rows", b, "logp err", (logp - ref)[b], "dbg", dbg[b], "dbg ref", dref[b])
