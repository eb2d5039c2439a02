import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import test_amortized as ta
g = ta.load("fa_e1e2_hw4")
print(g["meta"]["pdf_defs"], g["meta"]["flow_defs"], g["meta"]["kwargs"])
pdf = ta.build(g, torch.float64, "cuda")
cond = torch.from_numpy(g["cond"]).cuda()
amort = pdf.amortization_mlp(cond)
z = torch.from_numpy(g["z"]).cuda()
sx, _, slogp, _ = pdf.pdf_to_amortize._obtain_sample(predefined_target_input=z, amortization_parameters=amort)
err = np.abs(sx.cpu().numpy() - g["sample_x"])
bad = np.where(err.max(axis=1) > 1e-6)[0]
print("status", pdf.pdf_to_amortize.last_status_words, "bad rows", bad)
for r in bad[:4]:
    print(r, "z", g["z"][r], "got", sx[r].cpu().numpy(), "ref", g["sample_x"][r])
