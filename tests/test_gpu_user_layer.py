"""A USER-WRITTEN layer through the drop-in boundary: a plain-torch `euclidean_base` subclass that brings only
`_flow_mapping` / `_inv_flow_mapping` (+ the bookkeeping hooks), registered in `flow_options.opts_dict` the way the reference's own layers are
(jammy_flows/flow_options.py:25-240, jammy_flows/layers/layer_base.py:58-70), runs through `pdf` in both directions and under autograd -- next
to the library's kernel-backed layers in the same pdf -- and its log-probability is its analytic one."""
import math
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import jammy_flows_amd as jf
from jammy_flows_amd import flow_options
from jammy_flows_amd.layers.euclidean import euclidean_base

pytestmark = pytest.mark.gpu
HALF_LN_2PI = 0.5 * math.log(2.0 * math.pi)


class user_affine(euclidean_base.euclidean_base):
    """y = s * x + t per coordinate, s = floor + softplus(raw): parameters (raw_1..D, t_1..D).  Pure torch: no kernel of the library is involved."""

    def __init__(self, dimension, use_permanent_parameters=False, model_offset=0, scale_floor=0.1):
        super().__init__(dimension=dimension, use_permanent_parameters=use_permanent_parameters, model_offset=model_offset)
        self.scale_floor = scale_floor
        self.total_param_num += 2 * dimension
        if use_permanent_parameters:
            self.raw = torch.nn.Parameter(torch.randn(1, 2 * dimension) * 0.3)

    def _p(self, x, extra_inputs):
        p = self.raw.to(x) if extra_inputs is None else extra_inputs
        return self.scale_floor + F.softplus(p[:, :self.dimension]), p[:, self.dimension:]

    def _flow_mapping(self, inputs, extra_inputs=None):          # sampling direction
        x, log_det = inputs
        s, t = self._p(x, extra_inputs)
        return x * s + t, log_det + torch.log(s).sum(dim=1).expand(x.shape[0])

    def _inv_flow_mapping(self, inputs, extra_inputs=None):      # log-prob direction
        y, log_det = inputs
        s, t = self._p(y, extra_inputs)
        return (y - t) / s, log_det - torch.log(s).sum(dim=1).expand(y.shape[0])

    def _get_desired_init_parameters(self):
        return torch.cat([torch.zeros(self.dimension), torch.zeros(self.dimension)])

    def _init_params(self, params):
        self.raw.data = params.reshape(1, -1).to(self.raw.data)

    def _obtain_layer_param_structure(self, param_dict, extra_inputs=None, previous_x=None, extra_prefix=""):
        param_dict[extra_prefix + "raw"] = self.raw.data if extra_inputs is None else extra_inputs


@pytest.fixture(scope="module", autouse=True)
def registered():
    flow_options.opts_dict["q"] = {"module_path": __name__, "class_name": "user_affine", "type": "e",
                                   "kwargs": {"skip_model_offset": (0, [0, 1]), "scale_floor": (0.1, lambda v: v > 0)}}
    sys.modules.setdefault(__name__, sys.modules[__name__])
    yield
    flow_options.opts_dict.pop("q", None)


def analytic_logp(layer, offsets, x):
    s = layer.scale_floor + F.softplus(layer.raw[:, :layer.dimension].to(x))
    t = layer.raw[:, layer.dimension:].to(x)
    z = (x - offsets.to(x).reshape(1, -1) - t) / s
    return (-0.5 * z * z - HALF_LN_2PI).sum(1) - torch.log(s).sum(), z


def test_user_layer_alone_both_directions_and_autograd():
    torch.manual_seed(3)
    pdf = jf.pdf("e3", "q", options_overwrite={"q": {"scale_floor": 0.2}}).double().cuda()
    layer = pdf.layer_list[0][0]
    assert type(layer) is user_affine and layer.scale_floor == 0.2 and layer.model_offset == 1
    with torch.no_grad():
        layer.raw.normal_(0, 0.5)
        layer.offsets.normal_(0, 0.5)
    x = torch.randn(500, 3, dtype=torch.float64, device="cuda")
    x_before = x.clone()
    with torch.no_grad():
        logp, logp_base, base = pdf(x)
        want, z = analytic_logp(layer, layer.offsets, x)
    assert torch.equal(x, x_before)
    assert torch.allclose(logp, want, rtol=0, atol=1e-12) and torch.allclose(base, z, rtol=0, atol=1e-12)
    # sampling direction: x = flow(z), and the log-prob of the sample is the one returned with it
    with torch.no_grad():
        xs, _, slogp, _ = pdf._obtain_sample(predefined_target_input=z)
        assert torch.allclose(xs, x, rtol=0, atol=1e-12) and torch.allclose(slogp, want, rtol=0, atol=1e-11)
        smp, _, lp, _ = pdf.sample(samplesize=64)
        assert torch.allclose(pdf(smp)[0], lp, rtol=0, atol=1e-10)
    # autograd through pdf.forward = autograd through the analytic expression  (the suite runs with gradients off by default: conftest.py)
    with torch.enable_grad():
        loss = -pdf(x)[0].mean()
        g = torch.autograd.grad(loss, [layer.raw, layer.offsets])
        want_loss = -analytic_logp(layer, layer.offsets, x)[0].mean()
        gw = torch.autograd.grad(want_loss, [layer.raw, layer.offsets])
    for a, b in zip(g, gw):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)


def test_user_layer_next_to_kernel_layers_conditional_and_amortised():
    """pdf("e2+e2", "g+qg"): block 0 is the library's broadcast g chain, block 1 = a g layer followed by the user layer, both amortised by the
    block's MLP on (conditional input, x_0).  The user layer's slice of the MLP output arrives as its extra_inputs; removing the layer's own
    transformation by hand from the targets must give the log-prob of the same pdf evaluated with an identity in its place."""
    torch.manual_seed(5)
    pdf = jf.pdf("e2+e2", "g+gq", conditional_input_dim=3).double().cuda()
    blk = list(pdf.layer_list[1])
    assert [type(l).__name__ for l in blk] == ["gf_block", "user_affine"]
    x = torch.randn(400, 4, dtype=torch.float64, device="cuda")
    c = torch.randn(400, 3, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        logp, _, base = pdf(x, conditional_input=c)
        assert torch.isfinite(logp).all()
        # the MLP's output row: [g layer | user layer (offset D, raw D, t D)]; undo the user layer (the LAST layer, applied first in the log-prob direction)
        mlp_in = torch.cat([c, x[:, :2]], dim=1)
        params = pdf.mlp_predictors[1](mlp_in)
        q = blk[1]
        own = params[:, params.shape[1] - q.total_param_num:]
        off, raw = own[:, :2], own[:, 2:]
        s = q.scale_floor + F.softplus(raw[:, :2])
        y = (x[:, 2:] - off - raw[:, 2:]) / s
        ld_q = -torch.log(s).sum(1)
        # the g layer alone on y with its slice of the row, through the plugin API
        g_params = params[:, :blk[0].total_param_num]
        z1, ld1 = blk[0].inv_flow_mapping([y, ld_q.clone()], extra_inputs=g_params)[:2]
        p0 = pdf.mlp_predictors[0](c)                                # block 0 is amortised by the conditional input alone
        z0, ld0 = pdf.layer_list[0][0].inv_flow_mapping([x[:, :2].contiguous(), torch.zeros(400, dtype=torch.float64, device="cuda")], extra_inputs=p0)[:2]
        want = ld0 + ld1 + (-0.5 * torch.cat([z0, z1], 1) ** 2 - HALF_LN_2PI).sum(1)
    assert torch.allclose(logp, want, rtol=0, atol=1e-10)
    assert torch.allclose(base, torch.cat([z0, z1], 1), rtol=0, atol=1e-10)
    # sampling round trip and gradients reach the MLP that amortises the user layer
    with torch.no_grad():
        xs, _, slogp, _ = pdf._obtain_sample(conditional_input=c, predefined_target_input=base)
        assert torch.allclose(xs, x, rtol=0, atol=1e-7) and torch.allclose(slogp, logp, rtol=0, atol=1e-7)
    with torch.enable_grad():
        loss = -pdf(x, conditional_input=c)[0].mean()
        loss.backward()
    last = pdf.mlp_predictors[1][2]
    gq = last.weight.grad[last.weight.shape[0] - q.total_param_num:]
    assert torch.isfinite(gq).all() and float(gq.abs().max()) > 0
    # finite-difference check of one weight of the user layer's slice
    i, j = last.weight.shape[0] - 1, 5
    eps = 1e-6
    with torch.no_grad():
        w0 = float(last.weight[i, j])
        last.weight[i, j] = w0 + eps
        lp = float(-pdf(x, conditional_input=c)[0].mean())
        last.weight[i, j] = w0 - eps
        lm = float(-pdf(x, conditional_input=c)[0].mean())
        last.weight[i, j] = w0
    assert abs((lp - lm) / (2 * eps) - float(last.weight.grad[i, j])) < 1e-6


def test_kernel_caps_raise_loudly_on_the_device():
    """a g layer of 65 dimensions constructs (the reference has no cap, flow_options.py:38) but has no kernel (a wave = 64 lanes per row since
    round 6, 32 before): the forward call says so; 33 dimensions run"""
    pdf = jf.pdf("e65", "g").cuda()
    with pytest.raises((NotImplementedError, RuntimeError), match="65|dimension|unsupported|not supported"):
        pdf(torch.randn(10, 65, device="cuda"))
    pdf = jf.pdf("e33", "g").cuda()
    assert torch.isfinite(pdf(torch.randn(10, 33, device="cuda"))[0]).all()
