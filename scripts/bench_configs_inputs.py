"""SURVEY 8d synthetic inputs for any fixture's pdf definition (shared by scripts/bench_configs.py, scripts/rows_sweep.py, bench.py)."""
import numpy as np


def inputs(fx, n, seed):
    rng = np.random.default_rng(seed)
    cols = []
    for part in fx.pdf_defs.split("+"):
        kind, dim = part[0], int(part[1:].split("_")[0])
        if kind == "e":
            cols.append(rng.normal(size=(n, dim)) * 1.5)
        elif kind == "i":
            cols.append(rng.uniform(1e-6, 1 - 1e-6, size=(n, 1)))
        elif dim == 1:
            cols.append(rng.uniform(0, 2 * np.pi, size=(n, 1)))
        else:
            cols.append(np.arccos(rng.uniform(-1, 1, size=(n, 1))).clip(1e-3, np.pi - 1e-3))
            cols.append(rng.uniform(0, 2 * np.pi, size=(n, 1)))
    x = np.concatenate(cols, axis=1)
    c = fx.get("cond")
    cond = rng.normal(size=(n, c.shape[1])) if c is not None else None
    return x, cond
