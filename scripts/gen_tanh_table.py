#!/usr/bin/env python3
"""writes jammy_flows_amd/csrc/jf_tanh_table.h: tanh(i / 32), i = 0 .. 608, the lookup table of jf::tanh_tab (float64 hidden layers)"""
import math
import os

N = 609
vals = [math.tanh(i / 32.0) for i in range(N)]
lines = ["    " + ", ".join(repr(v) for v in vals[i:i + 4]) + "," for i in range(0, N, 4)]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jammy_flows_amd", "csrc", "jf_tanh_table.h")
open(out, "w").write("""// tanh(i / 32), i = 0 .. 608 (generated: python3 math.tanh, < 1 ulp; scripts/gen_tanh_table.py) -- the table of jf::tanh_tab (jf_math.h)
#pragma once
namespace jf {
constexpr int JF_TANH_TAB_N = 609;
static __device__ const double JF_TANH_TAB[JF_TANH_TAB_N] = {
""" + "\n".join(lines) + """
};
}  // namespace jf
""")
