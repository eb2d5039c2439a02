"""coefficients of M<double>::exp_fast (csrc/jf_math.h): degree-9 Chebyshev interpolant of e^r on |r| <= ln2 / 2, converted to the power basis"""
import numpy as np
from numpy.polynomial import chebyshev as C
h = np.log(2) / 2 * 1.0001
for deg in (8, 9, 10):
    co = C.cheb2poly(C.chebinterpolate(lambda t: np.exp(t * h), deg)) / h ** np.arange(deg + 1)
    r = np.linspace(-h, h, 200001)
    acc = np.full_like(r, co[-1])
    for c in co[-2::-1]:
        acc = acc * r + c
    ref = np.exp(r.astype(np.longdouble))
    print(deg, "max relative error %.2e" % float(np.max(np.abs(acc.astype(np.longdouble) - ref) / ref)))
    if deg == 9:
        print(", ".join("%.17e" % c for c in co))
