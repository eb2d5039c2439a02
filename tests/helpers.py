"""Shared helpers of the test-suite: build the product pdf / the oracle from a golden fixture."""
import numpy as np
import torch

import fixture_io


def product_supports(fx):
    """can jammy_flows_amd construct this fixture's pdf (i.e. do all its layers / options have kernels)?"""
    import jammy_flows_amd
    try:
        jammy_flows_amd.pdf(fx.pdf_defs, fx.flow_defs, **fx.kwargs)
        return True
    except NotImplementedError:
        return False


def build_product(fx, dtype, device="cuda"):
    import jammy_flows_amd
    torch.manual_seed(0)
    pdf = jammy_flows_amd.pdf(fx.pdf_defs, fx.flow_defs, **fx.kwargs).double()    # load at full precision, cast afterwards
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in fx.state_dict().items()}
    missing, unexpected = pdf.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    pdf = pdf.to(dtype=dtype, device=device)
    return pdf


def build_oracle(fx):
    from oracle import OraclePdf
    return OraclePdf(fx.pdf_defs, fx.flow_defs, state_dict=fx.state_dict(), **fx.kwargs)


def to_dev(arr, dtype, device="cuda"):
    return None if arr is None else torch.from_numpy(np.ascontiguousarray(arr)).to(dtype=dtype, device=device)


def max_abs(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def max_rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b)))) if a.size else 0.0


ALL_FIXTURES = [fixture_io.Fixture(p) for p in fixture_io.list_fixtures()]
