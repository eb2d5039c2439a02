"""does a HIP-graph replay of the training step reproduce the eager step?  (forward only, forward + backward, + Adam)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import numpy as np
import torch
import fixture_io, helpers

fx = fixture_io.load("c3_e4s2e4")
pdf = helpers.build_product(fx, torch.float32)
pdf.check_status = False
rng = np.random.default_rng(7)
n = 1 << 14
x = np.concatenate([rng.normal(size=(n, 4)) * 1.5, np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3),
                    rng.uniform(0, 2 * np.pi, size=(n, 1)), rng.normal(size=(n, 4)) * 1.5], axis=1)
x = torch.from_numpy(x).to(device="cuda", dtype=torch.float32)


def fwd():
    return pdf(x)[0]


def fwd_bwd():
    for p in pdf.parameters():
        p.grad = None
    logp = pdf(x)[0]
    loss = -logp.mean()
    loss.backward()
    return loss


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


with torch.no_grad():
    ref = fwd().clone()
    g, out = capture(fwd)
    g.replay()
    torch.cuda.synchronize()
    print("no-grad forward: max |graph - eager| =", (out - ref).abs().max().item())
ref = fwd_bwd().item()
gref = {k: p.grad.clone() for k, p in pdf.named_parameters() if p.grad is not None}
g, out = capture(fwd_bwd)
g.replay()
torch.cuda.synchronize()
print("loss eager %.6f graph %.6f" % (ref, out.item()))
for k, p in pdf.named_parameters():
    if p.grad is not None:
        d = (p.grad - gref[k]).abs().max().item()
        if d > 1e-6 * gref[k].abs().max().item():
            print("  grad differs:", k, d, gref[k].abs().max().item())

# ---- whole step with Adam (capturable): 6 replays against 6 eager steps from the same start
def run(graphed):
    pdf2 = helpers.build_product(fx, torch.float32)
    pdf2.check_status = False
    opt = torch.optim.Adam(pdf2.parameters(), lr=1e-2, capturable=True)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = -pdf2(x)[0].mean()
        loss.backward()
        opt.step()
        return loss

    losses = []
    if not graphed:
        for _ in range(9):
            losses.append(step().item())
        return losses
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            losses.append(step().item())
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        out = step()
    for _ in range(6):
        g.replay()
        losses.append(out.item())
    return losses


print("eager  ", ["%.4f" % v for v in run(False)])
print("graphed", ["%.4f" % v for v in run(True)])
