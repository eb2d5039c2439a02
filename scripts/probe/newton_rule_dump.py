"""sampling direction of every golden fixture in float64 with the library this process loads (JF_NEWTON_RULE=reference: libjammy_hip_audit.so, the
reference's own solver iteration) -> an .npz of samples, log-probs and Newton row-step counts.  tests/test_gpu_parity.py runs it as a child process
under the audit rule and compares with the product library's results.   python3 scripts/probe/newton_rule_dump.py out.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import fixture_io, helpers
from jammy_flows_amd import _hip

out = {"rule": np.array(_hip.get_newton_rule())}
for f in sorted(os.listdir(fixture_io.GOLDEN_DIR)):
    if not f.endswith(".npz"):
        continue
    fx = fixture_io.load(f[:-4])
    if not helpers.product_supports(fx):
        continue
    pdf = helpers.build_product(fx, torch.float64, "cuda")
    z = helpers.to_dev(fx["z"], torch.float64, "cuda")
    cond = helpers.to_dev(fx.get("cond"), torch.float64, "cuda")
    x, _, logp, _ = pdf._obtain_sample(conditional_input=cond, predefined_target_input=z, force_embedding_coordinates=fx.meta["embedding"])
    out[fx.name + "/x"] = x.cpu().numpy()
    out[fx.name + "/logp"] = logp.cpu().numpy()
    out[fx.name + "/steps"] = np.array(pdf.last_status_words["newton_row_steps"])
np.savez(sys.argv[1], **out)
print("fixtures", (len(out) - 1) // 3, "rule", _hip.get_newton_rule())
