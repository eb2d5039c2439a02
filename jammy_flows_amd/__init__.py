"""jammy_flows_amd -- MI355X-native (gfx950 / CDNA4) implementation of the per-layer forward / inverse + log-det hot path of
thoglu/jammy_flows behind the reference's own user API (``pdf("e4+s2+e4", "gggg+f+gggg")``, layer_base plugin classes).

All arithmetic runs in hand-written HIP kernels (``libjammy_hip.so``, C ABI in include/jammy_hip.h); there is no CPU fallback.
"""
from .main.default import pdf  # noqa: F401
from .main.fully_amortized import fully_amortized_pdf  # noqa: F401

__version__ = "0.1.0"
