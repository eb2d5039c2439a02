import os, sys
ROOT="/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np, torch, bench, fixture_io, helpers
torch.set_grad_enabled(False)
N=1<<15
for wl in ("c2",):
    W=bench.WORKLOADS[wl]; fx=fixture_io.load(W["fixture"])
    x,c=bench.make_inputs(wl,N,W["seed"])
    orc=helpers.build_oracle(fx)
    ref,_,rbase=orc.forward(x,c)
    p32=helpers.build_product(fx,torch.float32,"cuda"); p32.check_status=False
    p64=helpers.build_product(fx,torch.float64,"cuda"); p64.check_status=False
    xt=torch.from_numpy(x).cuda()
    l32,_,b32=p32(xt.float()); l64,_,b64=p64(xt.float().double())
    e=np.abs(l32.double().cpu().numpy()-l64.cpu().numpy())
    idx=np.argsort(-e)[:12]
    print("worst rows (f32 kernel vs f64 kernel on the same f32-rounded inputs):")
    for i in idx:
        print(i, "err %.2e"%e[i], "logp %.3f"%ref[i], "x", np.round(x[i],3), "base64", np.round(b64[i].cpu().numpy(),3), "dbase", (b32[i].double()-b64[i]).cpu().numpy())
    # per-layer: run the layers one at a time in f32 vs f64 from the f64 intermediate
    from jammy_flows_amd import _hip
    print("corr(err, max|base|):", np.corrcoef(e, np.abs(b64.cpu().numpy()).max(1))[0,1], " corr(err,|logp|):", np.corrcoef(e,np.abs(ref))[0,1])
    for thr in (2,3,4,5):
        m=np.abs(b64.cpu().numpy()).max(1)<thr
        print("rows with max|base|<%d: %d, max err %.2e mean %.2e"%(thr,m.sum(),e[m].max(),e[m].mean()))
