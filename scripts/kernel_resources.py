#!/usr/bin/env python3
"""VGPRs / AGPRs / spills / LDS / scratch of every gfx950 kernel inside libjammy_hip.so (from the code objects' metadata notes).
    python3 scripts/kernel_resources.py [name filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = data.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(magic))
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield data[pos + off:pos + off + size]
        pos = data.find(magic, pos + len(magic))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = os.path.join(ROOT, "jammy_flows_amd", "libjammy_hip.so")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, co in enumerate(code_objects(lib)):
            p = os.path.join(tmp, "d%d.co" % i)
            open(p, "wb").write(co)
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", p], stdout=subprocess.PIPE, check=True).stdout.decode()
            for blk in notes.split("- .agpr_count:")[1:]:
                get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", get("name")], stdout=subprocess.PIPE).stdout.decode().strip()
                name = re.sub(r"\(.*", "", name).replace("void jf::", "")
                rows.append((name, re.match(r"\s*(\d+)", blk).group(1), get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                             get("group_segment_fixed_size"), get("private_segment_fixed_size"), get("max_flat_workgroup_size")))
    print("%-110s %5s %5s %6s %6s %7s %8s %6s" % ("kernel", "agpr", "vgpr", "vspill", "sspill", "lds", "scratch", "wg"))
    for r in sorted(rows):
        if flt in r[0]:
            print("%-110s %5s %5s %6s %6s %7s %8s %6s" % ((r[0][:110],) + r[1:]))


if __name__ == "__main__":
    main()
