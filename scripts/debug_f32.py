import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import fixture_io, helpers
np.set_printoptions(linewidth=200, precision=6)
for name in sys.argv[1:]:
    fx = fixture_io.load(name)
    pdf = helpers.build_product(fx, torch.float32); pdf.check_status = False
    x = helpers.to_dev(fx["x"], torch.float32); cond = helpers.to_dev(fx.get("cond"), torch.float32)
    logp, lpb, base = pdf(x, conditional_input=cond, force_embedding_coordinates=fx.meta["embedding"])
    lp = logp.double().cpu().numpy(); err = np.abs(lp - fx["logp"])
    bad = np.nonzero(~np.isfinite(lp) | (err > 1e-2))[0]
    print(name, "bad rows", bad, "\n  got", lp[bad], "\n  ref", fx["logp"][bad], "\n  x", fx["x"][bad].tolist())
    good = np.isfinite(lp)
    print("  max err over finite rows %.3e ; median %.3e" % (err[good].max(), np.median(err[good])))
