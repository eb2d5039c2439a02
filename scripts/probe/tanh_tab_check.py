#!/usr/bin/env python3
"""numpy emulation of jf::tanh_tab (csrc/jf_math.h) against a 40-digit tanh: absolute / relative error over [-20, 20] and around 0"""
import math
import mpmath
import numpy as np

mpmath.mp.dps = 40
tab = np.array([math.tanh(i / 32.0) for i in range(609)])


def tanh_tab(x):
    ax = np.minimum(np.abs(x), 19.0)
    k = np.rint(ax * 32.0)
    r = ax - k * 0.03125
    T = tab[k.astype(int)]
    r2 = r * r
    p = r * (1.0 + r2 * (-1.0 / 3.0 + r2 * (2.0 / 15.0 + r2 * (-17.0 / 315.0))))
    return np.copysign((T + p) / (1.0 + T * p), x)


rng = np.random.default_rng(0)
xs = np.concatenate([rng.uniform(-20, 20, 20000), rng.normal(0, 1, 20000), rng.normal(0, 1e-3, 2000),
                     np.array([0.0, 1e-300, -1e-10, 19.0, 19.5, 40.0, -700.0, 1 / 64, 3 / 64, 0.5 + 1 / 64])])
got = tanh_tab(xs)
ref = [mpmath.tanh(mpmath.mpf(float(x))) for x in xs]
print("max abs err", max(abs(mpmath.mpf(float(g)) - t) for g, t in zip(got, ref)))
print("max rel err", max(abs((mpmath.mpf(float(g)) - t) / t) for g, t in zip(got, ref) if t != 0))
