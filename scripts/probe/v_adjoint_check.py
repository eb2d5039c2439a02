"""Closed-form adjoint of the 'v' potential (jf_expmap.h: v_component_adjoint; spline potentials: manifold_bwd_kernels.hip v_spline_adjoint)
against the dual-number replay of the same kernel (JF_V_BWD_DUAL=1): every potential kind, with / without rotation, permanent and conditional parameters.
Usage: python scripts/probe/v_adjoint_check.py            (runs itself twice as child processes and compares)"""
import os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]

CASES = [("s2", "v", k, rot, cond, nc) for k in ("exponential", "linear", "quadratic") for rot in (0, 1) for cond in (0, 2) for nc in (1, 10)]
CASES += [("s2", "v", "splines", rot, cond, nc) for rot in (0, 1) for cond in (0, 2) for nc in (1, 5)]      # round 6: v_spline_adjoint (seven-tangent bin evaluation + table reverse)
CASES += [("s2", "vv", "splines", 1, 2, 3)]
# round 6: natural_direction = 1 (the log-prob direction solves the exponential map: implicit-function adjoint in the staged kernel); 7th entry
CASES += [("s2", "v", k, rot, cond, 4, 1) for k in ("exponential", "linear", "quadratic", "splines") for rot in (0, 1) for cond in (0, 2)]
CASES += [("s2", "vv", "exponential", 1, 2, 3, 1)]
CASES += [("s2", "vv", "exponential", 1, 2, 4), ("e2+s2", "gg+v", "quadratic", 0, 3, 10),
          ("s2", "v", "exponential", 1, 2, 70)]          # 362 parameters per row: the kernel takes 32 rows per wave (LDS), half its lanes idle


def grads(out):
    import torch
    import jammy_flows_amd as jf
    res = {}
    for ci, case in enumerate(CASES):
        pd, fl, kind, rot, cond, nc = case[:6]
        nat = case[6] if len(case) > 6 else 0
        torch.manual_seed(ci)
        kw = {"options_overwrite": {"v": {"exp_map_type": kind, "add_rotation": rot, "num_components": nc, "natural_direction": nat}}}
        if cond:
            kw["conditional_input_dim"] = cond
        pdf = jf.pdf(pd, fl, **kw).double().cuda()
        g = torch.Generator().manual_seed(100 + ci)
        with torch.no_grad():
            for prm in pdf.layer_list.parameters():
                prm.add_(0.3 * torch.randn(prm.shape, generator=g, dtype=prm.dtype).cuda())
            for mlp in pdf.mlp_predictors:
                if mlp is None:
                    continue
                for m in mlp:
                    if hasattr(m, "weight"):
                        m.weight.mul_(200.0)
        B = 777
        cols = []
        for sub in pd.split("+"):
            if sub[0] == "e":
                cols.append(torch.randn(B, int(sub[1:]), generator=g, dtype=torch.float64))
            else:
                cols += [torch.rand(B, 1, generator=g, dtype=torch.float64) * 2.8 + 0.15, torch.rand(B, 1, generator=g, dtype=torch.float64) * 6.0 + 0.1]
        x = torch.cat(cols, 1).cuda().requires_grad_(True)
        c = torch.randn(B, cond, generator=g, dtype=torch.float64).cuda().requires_grad_(True) if cond else None
        with torch.enable_grad():
            lp = pdf(x, conditional_input=c)[0]
            (lp * torch.linspace(0.5, 1.5, B, dtype=torch.float64, device="cuda")).sum().backward()
        res["%d/x" % ci] = x.grad.cpu().numpy()
        if c is not None:
            res["%d/c" % ci] = c.grad.cpu().numpy()
        for n, p in pdf.named_parameters():
            if p.grad is not None:
                res["%d/%s" % (ci, n)] = p.grad.cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        grads(sys.argv[1])
        sys.exit(0)
    d = tempfile.mkdtemp()
    outs = []
    for dual in ("0", "1"):
        o = os.path.join(d, "g%s.npz" % dual)
        subprocess.run([sys.executable, __file__, o], check=True, env=dict(os.environ, JF_V_BWD_DUAL=dual))
        outs.append(np.load(o))
    worst, worst_nat1 = 0.0, 0.0
    for k in outs[0].files:
        a, b = outs[0][k], outs[1][k]
        e = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-9))
        nat1 = len(CASES[int(k.split("/")[0])]) > 6
        if nat1:                                        # the replay differentiates the Newton iteration step by step (as the reference does): 1e-4 is its own bar
            worst_nat1 = max(worst_nat1, e)
            if e > 1e-4:
                print("MISMATCH (natural_direction = 1)", k, CASES[int(k.split("/")[0])], e)
            continue
        worst = max(worst, e)
        if e > 1e-10:
            print("MISMATCH", k, CASES[int(k.split("/")[0])], e)
    print("cases %d, tensors %d, worst relative difference closed form vs dual replay: %.3e" % (len(CASES), len(outs[0].files), worst))
    print("natural_direction = 1 (implicit-function adjoint vs the replay of the Newton iteration): worst relative difference %.3e" % worst_nat1)
