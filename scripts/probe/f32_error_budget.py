"""float32 error budget: max |d log p| against the float64 oracle of C2 (unconditional g chain: no MLP) and C3 (the headline: fused
f16-split MLP + g layers) with ONE hardware approximation of csrc/jf_math.h at a time replaced by its correctly rounded library function
(probe libraries: scripts/probe/f32_error_budget.sh), and of the C3 block on exact-f32 MFMA instead of the 2-piece f16 split.
python3 scripts/probe/f32_error_budget.py            (GPU box; runs itself once per library as a child process: the library is loaded once)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
N = 1 << 15
VARIANTS = [("shipped", None), ("base (probe tree, no switch)", "base"), ("accurate exp (expf for v_exp_f32)", "exp"), ("accurate log (logf for v_log_f32)", "log"),
            ("accurate rcp / sqrt (IEEE division for v_rcp_f32)", "rcp"), ("all three accurate", "all")]


def oracle_cache():
    import numpy as np
    import bench
    import fixture_io, helpers
    path = os.path.join("/tmp", "jf_f32_budget_oracle.npz")
    if os.path.exists(path):
        return dict(np.load(path))
    out = {}
    for wl in ("c2", "c3"):
        W = bench.WORKLOADS[wl]
        x, c = bench.make_inputs(wl, N, W["seed"])
        out[wl + "_x"] = x
        out[wl + "_logp"] = helpers.build_oracle(fixture_io.load(W["fixture"])).forward(x, c)[0]
    np.savez(path, **out)
    return out


def child():
    import numpy as np
    import torch
    import bench
    import fixture_io, helpers
    torch.set_grad_enabled(False)
    o = oracle_cache()
    res = {}
    for wl in ("c2", "c3"):
        fx = fixture_io.load(bench.WORKLOADS[wl]["fixture"])
        x = torch.from_numpy(o[wl + "_x"]).to(device="cuda", dtype=torch.float32)
        for arith in (["split", "f32"] if wl == "c3" else ["-"]):
            pdf = helpers.build_product(fx, torch.float32, "cuda")
            pdf.check_status = False
            if arith == "f32":
                pdf.fused_matrix_arithmetic = "f32"              # the 128 -> 548 product on exact-f32 MFMA instead of three f16 passes over 2-piece splits
            lp = pdf(x)[0].double().cpu().numpy()
            ref = o[wl + "_logp"]
            fin = np.isfinite(ref) & (np.abs(ref) < 1e4)
            err = np.abs(lp - ref)[fin]
            res["%s%s" % (wl, "" if arith == "-" else "/" + arith)] = {"max": float(err.max()), "p99.9": float(np.quantile(err, 0.999)), "mean": float(err.mean())}
    if os.environ.get("JF_LIB_PATH") is None:
        # the representation floor: FLOAT64 arithmetic (the float64 kernels) on what a float32 run is given -- the inputs rounded to float32, and
        # the inputs and every parameter rounded to float32 -- against the same oracle values
        for wl in ("c2", "c3"):
            fx = fixture_io.load(bench.WORKLOADS[wl]["fixture"])
            ref = o[wl + "_logp"]
            fin = np.isfinite(ref) & (np.abs(ref) < 1e4)
            x32 = torch.from_numpy(o[wl + "_x"]).to(device="cuda", dtype=torch.float32).double()
            for tag, round_params in (("f64 arithmetic, float32-rounded inputs", False), ("f64 arithmetic, float32-rounded inputs and parameters", True)):
                pdf = helpers.build_product(fx, torch.float64, "cuda")
                pdf.check_status = False
                if round_params:
                    for prm in pdf.parameters():
                        prm.data = prm.data.float().double()
                    pdf.invalidate_packed_caches() if hasattr(pdf, "invalidate_packed_caches") else None
                err = np.abs(pdf(x32)[0].cpu().numpy() - ref)[fin]
                res["%s|%s" % (wl, tag)] = {"max": float(err.max()), "p99.9": float(np.quantile(err, 0.999)), "mean": float(err.mean())}
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    oracle_cache()
    rows = []
    for label, v in VARIANTS:
        env = dict(os.environ)
        if v is not None:
            lib = os.path.join(ROOT, "build_probe", "f32_budget", v, "libjammy_hip.so")
            if not os.path.exists(lib):
                print("(no probe library for %s: run scripts/probe/f32_error_budget.sh in the build container first)" % v)
                continue
            env["JF_LIB_PATH"] = lib
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(label, "FAILED", r.stderr[-500:])
            continue
        rows.append((label, json.loads(line[0][7:])))
    print("| library | C2 max / p99.9 / mean | C3 (f16 split) max / p99.9 / mean | C3 (exact f32 MFMA) max / p99.9 / mean |")
    print("|---|---|---|---|")
    for label, r in rows:
        f = lambda k: "%.2e / %.2e / %.2e" % (r[k]["max"], r[k]["p99.9"], r[k]["mean"]) if k in r else "-"
        print("| %s | %s | %s | %s |" % (label, f("c2"), f("c3/split"), f("c3/f32")))
    print("(max |d log p| against the float64 oracle over %d rows of the bench inputs, rows with |log p| < 1e4)" % N)
    floor = {k: v for k, v in rows[0][1].items() if "|" in k} if rows else {}
    if floor:
        print()
        print("| float64 kernels on float32-representable data | C2 max / p99.9 / mean | C3 max / p99.9 / mean |")
        print("|---|---|---|")
        for tag in sorted({k.split("|")[1] for k in floor}):
            g = lambda wl: "%.2e / %.2e / %.2e" % (floor[wl + "|" + tag]["max"], floor[wl + "|" + tag]["p99.9"], floor[wl + "|" + tag]["mean"])
            print("| %s | %s | %s |" % (tag, g("c2"), g("c3")))
