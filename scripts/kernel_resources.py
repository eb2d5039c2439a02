#!/usr/bin/env python3
"""registers / scratch / LDS of every kernel in the shipped libjammy_hip.so (from the code objects' metadata notes):
python3 scripts/kernel_resources.py [--scratch]      (--scratch: only kernels with a private segment)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from test_host_logic import _device_code_objects  # noqa: E402

lib = os.path.join(ROOT, "jammy_flows_amd", "libjammy_hip.so")
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
only = "--scratch" in sys.argv
rows = []
for co in _device_code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        txt = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count")[1:]:
        def g(k):
            m = re.search(r"\.%s:\s*(\S+)" % k, blk)
            return m.group(1) if m else "?"
        rows.append((g("name"), g("vgpr_count"), g("agpr_count") if False else blk.split()[0].lstrip(":"), g("private_segment_fixed_size"), g("vgpr_spill_count"),
                     g("sgpr_spill_count"), g("group_segment_fixed_size")))
for name, v, a, p, vs, ss, l in sorted(rows, key=lambda r: -int(r[3]) if r[3].isdigit() else 0):
    if only and (not p.isdigit() or int(p) == 0):
        continue
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%-110s vgpr %4s scratch %5s B  vgpr spills %3s sgpr spills %3s lds %6s" % (d[:110], v, p, vs, ss, l))
