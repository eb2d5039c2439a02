"""debug helper: per-row comparison of the HIP path with a golden fixture (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import fixture_io, helpers
name = sys.argv[1]
dt = torch.float64 if len(sys.argv) < 3 or sys.argv[2] == "f64" else torch.float32
fx = fixture_io.load(name)
pdf = helpers.build_product(fx, dt)
x = helpers.to_dev(fx["x"], dt); cond = helpers.to_dev(fx.get("cond"), dt)
logp, lpb, base = pdf(x, conditional_input=cond, force_embedding_coordinates=fx.meta["embedding"])
err = np.abs(logp.double().cpu().numpy() - fx["logp"])
idx = np.argsort(-err)[:8]
print("worst rows", idx, "\nerr", err[idx], "\nlogp ref", fx["logp"][idx], "\nx", fx["x"][idx])
print("base err", np.abs(base.double().cpu().numpy() - fx["base"])[idx])
# per-layer comparison against the reference trace (single e-block pdfs only)
if len(pdf.layer_list) == 1:
    extra = None
    if pdf.mlp_predictors[0] is not None:
        extra = pdf.mlp_predictors[0](cond)
    cur, ld, used = x, torch.zeros(x.shape[0], dtype=dt, device="cuda"), 0
    tr = fx.trace("inv")
    for i, layer in enumerate(reversed(list(pdf.layer_list[0]))):
        this = None
        if extra is not None:
            end = extra.shape[1] - used
            this = extra[:, end - layer.total_param_num:end]
        cur, ld = layer.inv_flow_mapping([cur, ld], extra_inputs=this)
        used += layer.total_param_num
        ex = np.abs(cur.double().cpu().numpy() - tr[i][1]); el = np.abs(ld.double().cpu().numpy() - tr[i][2])
        print("layer", tr[i][0], "max x err %.3e (row %d)  max ld err %.3e (row %d)" % (ex.max(), ex.max(axis=1).argmax(), el.max(), el.argmax()), "median ld err %.3e" % np.median(el))
if len(pdf.layer_list) == 1 and pdf.mlp_predictors[0] is not None:
    orc = helpers.build_oracle(fx)
    oparams = orc.mlps[0](fx["cond"])
    print("MLP out max err vs numpy: %.3e" % np.abs(extra.double().cpu().numpy() - oparams).max())
