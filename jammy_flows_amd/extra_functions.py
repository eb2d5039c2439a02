"""Small host-side helpers (jammy_flows/extra_functions.py:81-95)."""
from torch import nn


def list_from_str(spec):
    """'64-30' -> [64, 30]; '' -> []."""
    if spec == "":
        return []
    return [int(s) for s in spec.split("-")]


# the names the reference offers (extra_functions.py:81-89); tanh is fused into the dense kernels, the others run as an elementwise launch on the
# pre-activation (jf_activation; "swish" with the reference's fixed beta = 1)
NONLINEARITIES = ("tanh", "relu", "softplus", "elu", "swish", "square", "identity")
