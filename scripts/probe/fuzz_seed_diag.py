"""one seed of scripts/probe/fuzz_more.py in detail: where the analytic x-gradient and finite differences disagree, at several step sizes and one-sided:
python3 scripts/probe/fuzz_seed_diag.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import test_gpu_fuzz as F

seed = int(sys.argv[1])
rng, pdf_defs, flow_defs, kwargs, pdf = F.build_fuzz_pdf(seed)
print(pdf_defs, flow_defs, kwargs)
pdf = pdf.cuda()
B = 48
x = torch.from_numpy(F.domain_rows(pdf_defs, B, rng)).cuda().requires_grad_(True)
c = torch.from_numpy(rng.normal(size=(B, 2))).cuda() if "conditional_input_dim" in kwargs else None
with torch.enable_grad():
    pdf(x, conditional_input=c)[0].sum().backward()
g = x.grad.clone()
with torch.no_grad():
    f0 = pdf(x.detach(), conditional_input=c)[0]
    worst = None
    for j in range(x.shape[1]):
        tp, tm = x.detach().clone(), x.detach().clone()
        tp[:, j] += 1e-6; tm[:, j] -= 1e-6
        fd = (pdf(tp, conditional_input=c)[0] - pdf(tm, conditional_input=c)[0]) / 2e-6
        e = ((g[:, j] - fd).abs() / (1 + fd.abs()))
        i = int(e.argmax())
        if worst is None or float(e[i]) > worst[0]:
            worst = (float(e[i]), i, j)
    err, i, j = worst
    print("worst row %d column %d: x = %s analytic %.8f, rel err %.2e" % (i, j, x[i].tolist(), float(g[i, j]), err))
    for eps in (1e-4, 1e-5, 1e-6, 1e-7):
        tp, tm = x.detach().clone(), x.detach().clone()
        tp[i, j] += eps; tm[i, j] -= eps
        fp, fm = pdf(tp, conditional_input=c)[0][i], pdf(tm, conditional_input=c)[0][i]
        print("eps %.0e: central %.8f  right %.8f  left %.8f" % (eps, float((fp - fm) / (2 * eps)), float((fp - f0[i]) / eps), float((f0[i] - fm) / eps)))
