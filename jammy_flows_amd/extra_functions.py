"""Small host-side helpers (jammy_flows/extra_functions.py:81-95)."""
from torch import nn


def list_from_str(spec):
    """'64-30' -> [64, 30]; '' -> []."""
    if spec == "":
        return []
    return [int(s) for s in spec.split("-")]


NONLINEARITIES = {"tanh": nn.Tanh()}
