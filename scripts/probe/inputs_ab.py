import sys, time
sys.path[:0]=[".","tests","tests/golden","scripts"]
import torch, numpy as np, fixture_io, helpers
from bench_configs_inputs import inputs
torch.set_grad_enabled(False)
for name in ("c2_e4_gggg","c3_e4s2e4"):
    fx=fixture_io.load(name)
    pdf=helpers.build_product(fx, torch.float32); pdf.check_status=False; pdf.use_step_plans=True
    n=1<<20
    xs=torch.from_numpy(inputs(fx,n,7)[0]).to(device="cuda",dtype=torch.float32)
    z=torch.randn(n, pdf.total_base_dim, device="cuda")
    xm=pdf._obtain_sample(predefined_target_input=z)[0].contiguous()
    for label,x in (("survey inputs",xs),("model samples",xm)):
        for _ in range(300): pdf(x)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(100): pdf(x)
        torch.cuda.synchronize(); print(name,label,"%.4f ms"%((time.perf_counter()-t0)/100*1e3))
