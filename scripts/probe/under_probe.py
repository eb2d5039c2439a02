import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "scripts")]
import torch, fixture_io, helpers
from bench_configs_inputs import inputs
from jammy_flows_amd import _hip
torch.set_grad_enabled(False)
for name in ("c2_e4_gggg", "c3_e4s2e4"):
    fx = fixture_io.load(name)
    pdf = helpers.build_product(fx, torch.float32)
    B = 1 << 20
    x_np, _ = inputs(fx, B, 7)
    x = torch.from_numpy(x_np).to(device="cuda", dtype=torch.float32)
    layers = list(pdf.layer_list[0])
    from jammy_flows_amd.layers.euclidean import gaussianization_flow as gfl
    params = gfl.chain_permanent_row(layers, x)
    larr = _hip.gf_layer_array([l.c_struct() for l in layers])
    st = _hip.new_status(x.device)
    _hip.gf_chain("inv", x[:, :4].contiguous(), None, params, larr, len(layers), 4, status=st)
    w = st.cpu().tolist()
    print(name, "under lanes (of %d coordinate-layer evaluations): %d = %.4f; waves re-evaluated: %d of %d = %.3f" % (B * 16, w[2], w[2] / (B * 16), w[3], B // 64 * 16, w[3] / (B // 64 * 16)))
    # model samples
    z = torch.randn(B, 4, device="cuda")
    xs, _ = _hip.gf_chain("fwd", z, None, params, larr, len(layers), 4)
    st = _hip.new_status(x.device)
    _hip.gf_chain("inv", xs, None, params, larr, len(layers), 4, status=st)
    w = st.cpu().tolist()
    print(name, "model samples: under lanes %d, waves %d" % (w[2], w[3]))
