#!/usr/bin/env python3
"""Golden vectors for the Moebius layer 'm' with use_moebius_xyz_parametrization=False (moebius_1d.py:39-46, 175-178: omega given by an angle;
three parameters per component) from the REAL reference.  flow_options.py does not expose the switch, so the layer class is instantiated
directly, with permanent parameters and with per-sample parameters (extra_inputs), natural_direction 0 and 1.

    cd /tmp && MPLBACKEND=Agg python /root/repo/tests/golden/make_moebius_angle_fixture.py
"""
import contextlib
import io
import os
import sys

import numpy
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
with contextlib.redirect_stdout(io.StringIO()):
    from jammy_flows.layers.spheres.moebius_1d import moebius  # noqa: E402

rng = numpy.random.default_rng(9)
B, nc = 96, 5
out = {"x": rng.uniform(0.01, 2 * numpy.pi - 0.01, size=(B, 1)), "extra": 0.5 * rng.normal(size=(B, nc * 3))}
for nd in (0, 1):
    torch.manual_seed(3 + nd)
    layer = moebius(dimension=1, euclidean_to_sphere_as_first=False, add_rotation=0, natural_direction=nd, use_permanent_parameters=True,
                    use_moebius_xyz_parametrization=False, num_basis_functions=nc).double()
    out["nd%d/pars" % nd] = layer.moebius_pars.detach().numpy().copy()
    for tag, extra in (("perm", None), ("cond", torch.from_numpy(out["extra"]))):
        x = torch.from_numpy(out["x"])
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            y, ld = layer.inv_flow_mapping([x.clone(), torch.zeros(B, dtype=torch.float64)], extra_inputs=extra)
            xs, lds = layer.flow_mapping([x.clone(), torch.zeros(B, dtype=torch.float64)], extra_inputs=extra)
        out["nd%d/%s/inv_y" % (nd, tag)] = y.numpy(); out["nd%d/%s/inv_ld" % (nd, tag)] = ld.numpy()
        out["nd%d/%s/fwd_x" % (nd, tag)] = xs.numpy(); out["nd%d/%s/fwd_ld" % (nd, tag)] = lds.numpy()
        print("nd", nd, tag, "inv ld range", float(ld.min()), float(ld.max()), "fwd ld range", float(lds.min()), float(lds.max()))
os.makedirs(os.path.join(HERE, "nonlin"), exist_ok=True)
path = os.path.join(HERE, "nonlin", "m_angle_layer.npz")
numpy.savez_compressed(path, **out)
print(os.path.getsize(path), "bytes ->", path)
