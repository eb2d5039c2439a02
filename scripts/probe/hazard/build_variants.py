#!/usr/bin/env python3
"""DESIGN.md 3.9 hunt, round 4: variants of cond_split_kernels.hip built WITH packed f32 instructions (the configuration that produced wrong
log-dets in whole 16-row groups) whose device assembly is post-processed before it is assembled -- wait states inserted at chosen places --
so that the hazard can be cornered by instruction class instead of by build flag.

    python3 scripts/probe/hazard/build_variants.py            (CPU box: hipcc cross-compiles; writes scripts/probe/hazard/out/cs_<variant>.o)
    bash scripts/probe/hazard/run_variants.sh                 (GPU box: links one library per variant, runs scripts/probe/rgcheck.py on each)

Pipeline per variant (the commands `hipcc -save-temps -###` prints, replayed by hand around the edited .s):
    device .s --edit--> cc1as -> lld -> clang-offload-bundler -> .hipfb;  host .s with the fat binary's .asciz replaced by .incbin -> cc1as -> .o
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
CSRC = os.path.join(ROOT, "jammy_flows_amd", "csrc")
OUT = os.path.join(HERE, "out")
LLVM = "/opt/rocm/lib/llvm/bin"
SRC = os.environ.get("HZ_SRC", "cond_split_kernels.hip")
STEM = SRC[:-4]

TRANS = ("v_log_f32", "v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_log_legacy_f32", "v_exp_legacy_f32",
         "v_rcp_iflag_f32")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
INSN = re.compile(r"^\t([a-z_0-9]+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(line):
    body = line.strip().split(None, 1)
    if len(body) < 2:
        return "", ""
    ops = body[1].split(";")[0]
    first, _, rest = ops.partition(",")
    return first, rest


def transform(lines, mode, n):
    """mode: none | before_pk | after_pk | after_trans | dep_pk (before a v_pk_* that reads a register a transcendental wrote within the last
    `window` instruction slots) | dep_any (the same for ANY VALU consumer) | pad_random (control: the same number of s_nop as before_pk, after v_fma)"""
    out, slot = [], 0
    trans_def = {}                # vreg -> slot index of the transcendental that wrote it last
    window = int(os.environ.get("HZ_WINDOW", "8"))
    n_ins = 0
    for line in lines:
        m = INSN.match(line)
        if not m:
            out.append(line)
            continue
        op = m.group(1)
        is_pk = op.startswith("v_pk_") and op.endswith("_f32")
        is_trans = op.startswith(TRANS)
        is_valu = op.startswith("v_") and not op.startswith(("v_mfma", "v_accvgpr"))
        first, rest = split_operands(line)
        nop = "\ts_nop %d\n" % (n - 1)
        if mode == "before_pk" and is_pk:
            out.append(nop); n_ins += 1; slot += n
        if mode in ("dep_pk", "dep_any") and ((is_pk and mode == "dep_pk") or (is_valu and mode == "dep_any")):
            srcs = regs_of(rest)
            if any(r in trans_def and slot - trans_def[r] <= window for r in srcs):
                out.append(nop); n_ins += 1; slot += n
        out.append(line)
        if op == "s_nop":
            slot += int(line.split()[1]) + 1
        else:
            slot += 1
        if is_valu:
            for r in regs_of(first):
                trans_def.pop(r, None)
            if is_trans:
                for r in regs_of(first):
                    trans_def[r] = slot
        if mode == "after_pk" and is_pk:
            out.append(nop); n_ins += 1; slot += n
        if mode == "after_trans" and is_trans:
            out.append(nop); n_ins += 1; slot += n
        if mode == "pad_fma" and op.startswith("v_fma_f32"):
            out.append(nop); n_ins += 1; slot += n
    return out, n_ins


PK = re.compile(r"^\t(v_pk_(mul|add|fma)_f32) (.*)$")
MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")
TARGET_KERNEL = os.environ.get("HZ_KERNEL", "_ZN2jf20cond_gf_split_kernelILi2ELb0ELb0ELi2EEEvNS_6CsArgsE")


def unpack_one(line):
    """v_pk_{mul,add,fma}_f32 -> its two single instructions (VOP3P semantics: the low result reads source half op_sel[i], the high result half
    op_sel_hi[i] (defaults 0 / 1), neg_lo / neg_hi negate per half), or None when the rewrite is not safe (overlap needing a temporary)."""
    m = PK.match(line.rstrip("\n"))
    if not m:
        return None
    op, kind, rest = m.group(1), m.group(2), m.group(3).split(";")[0].strip()
    mods = {k: [int(c) for c in v.split(",")] for k, v in MOD.findall(rest)}
    ops = [o.strip() for o in MOD.sub("", rest).strip().rstrip(",").split(",")]
    ops = [o for o in ops if o]
    nsrc = 3 if kind == "fma" else 2
    if len(ops) != nsrc + 1:
        return None
    dm = re.fullmatch(r"v\[(\d+):(\d+)\]", ops[0])
    if not dm:
        return None
    d0, d1 = int(dm.group(1)), int(dm.group(2))
    op_sel = mods.get("op_sel", [0] * nsrc) + [0] * nsrc
    op_sel_hi = mods.get("op_sel_hi", [1] * nsrc) + [1] * nsrc
    neg_lo = mods.get("neg_lo", [0] * nsrc) + [0] * nsrc
    neg_hi = mods.get("neg_hi", [0] * nsrc) + [0] * nsrc

    def half(src, sel, neg):
        rm = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", src)
        if rm:
            r = "%s%d" % (rm.group(1), int(rm.group(2)) + sel)
        else:
            if sel:                                   # the high half of an inline constant / single register: not a form seen here
                return None
            r = src
        return ("-" if neg else "") + r
    lo = [half(ops[1 + i], op_sel[i], neg_lo[i]) for i in range(nsrc)]
    hi = [half(ops[1 + i], op_sel_hi[i], neg_hi[i]) for i in range(nsrc)]
    if None in lo or None in hi:
        return None
    single = {"mul": "v_mul_f32_e64", "add": "v_add_f32_e64", "fma": "v_fma_f32"}[kind]
    a = "\t%s v%d, %s\n" % (single, d0, ", ".join(lo))
    b = "\t%s v%d, %s\n" % (single, d1, ", ".join(hi))
    lo_reads = {x.lstrip("-") for x in lo}
    hi_reads = {x.lstrip("-") for x in hi}
    if "v%d" % d0 not in hi_reads:
        return [a, b]
    if "v%d" % d1 not in lo_reads:
        return [b, a]
    # each half reads the other half's destination (an in-place swizzle such as v[68:69] = {v69 + v106, v68 + v107}): exchange the two
    # registers first (v_swap_b32), then every read of one names the other
    def ren(x):
        neg, r = ("-", x[1:]) if x.startswith("-") else ("", x)
        return neg + ("v%d" % d1 if r == "v%d" % d0 else "v%d" % d0 if r == "v%d" % d1 else r)
    lo2, hi2 = [ren(x) for x in lo], [ren(x) for x in hi]
    if "v%d" % d0 in {x.lstrip("-") for x in hi2}:
        return None
    return ["\tv_swap_b32 v%d, v%d\n" % (d0, d1), "\t%s v%d, %s\n" % (single, d0, ", ".join(lo2)), "\t%s v%d, %s\n" % (single, d1, ", ".join(hi2))]


def unpack(lines, keep_lo, keep_hi):
    """every packed f32 arithmetic instruction as two single ones, EXCEPT those with index in [keep_lo, keep_hi) inside TARGET_KERNEL"""
    out, cur, idx, n_un, n_kept, n_skip = [], None, 0, 0, 0, 0
    for line in lines:
        if line.startswith("_ZN") and ":" in line.split(";")[0]:
            cur = line.split(":")[0]
            idx = 0
        if PK.match(line.rstrip("\n")):
            in_target = cur == TARGET_KERNEL
            i = idx
            idx += 1
            if in_target and keep_lo <= i < keep_hi:
                out.append(line); n_kept += 1
                continue
            two = unpack_one(line)
            if two is None:
                out.append(line); n_skip += 1
                if in_target:
                    print("   not unpacked (target kernel, index %d): %s" % (i, line.strip()))
                continue
            out += two; n_un += 1
            continue
        out.append(line)
    return out, (n_un, n_kept, n_skip)


CTX_A = re.compile(r"^\tv_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] op_sel_hi:\[1,0\] neg_lo:\[0,1\] neg_hi:\[0,1\]\s*$")


def ctx(lines, which):
    """round-4 result of the bisect: ONE packed instruction form reintroduces the failure,
           A: v_pk_add_f32 v[a:b], v[c:d], v[e:f] op_sel:[0,1] op_sel_hi:[1,0] neg..     (a' = c - f,  b' = d - e)
              s_nop 0
              v_mov_b32 vb, vf                                                            (the high result of A is dead: overwritten at once)
           B: v_pk_add_f32 v[g:h], v[a:b], v[e:f] op_sel_hi:[1,0] neg..                   (g = a' - e,  h = f - e)
    Variants of that neighbourhood in the otherwise untouched packed build: unpackA | padAM (24 wait states between A and the v_mov) |
    padMB (between the v_mov and B) | padA (before A) | unpackB_late_mov (B as two single adds that do not read vb; the v_mov after them)"""
    out, n, i = [], 0, 0
    pad = ["\ts_nop 7\n"] * 3
    while i < len(lines):
        m = CTX_A.match(lines[i])
        ok = m and i + 3 < len(lines) and lines[i + 1].strip() == "s_nop 0" and lines[i + 2].startswith("\tv_mov_b32_e32 v%s, v%s" % (m.group(2), m.group(6)))
        if not ok:
            out.append(lines[i]); i += 1
            continue
        a, b, c, d, e, f = (int(t) for t in m.groups())
        A, NOP, MOV, B = lines[i:i + 4]
        n += 1
        if which == "unpackA":
            out += unpack_one(A) + [NOP, MOV, B]
        elif which == "padAM":
            out += [A] + pad + [MOV, B]
        elif which == "padMB":
            out += [A, NOP, MOV] + pad + [B]
        elif which == "padA":
            out += pad + [A, NOP, MOV, B]
        elif which == "commuteA":          # the same sum with the operands exchanged (the crossing then sits on src0)
            out += ["\tv_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]\n" % (a, b, e, f, c, d), NOP, MOV, B]
        elif which == "fmaA":              # the same values from v_pk_fma_f32: src1 * (-1.0) + src0, crossing on the multiplicand
            out += ["\tv_pk_fma_f32 v[%d:%d], v[%d:%d], -1.0, v[%d:%d] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n" % (a, b, e, f, c, d), NOP, MOV, B]
        elif which == "swapA":             # no crossing: the halves of src1 exchanged around a straight packed subtraction
            out += ["\tv_swap_b32 v%d, v%d\n" % (e, f), "\tv_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] neg_lo:[0,1] neg_hi:[0,1]\n" % (a, b, c, d, e, f),
                    "\tv_swap_b32 v%d, v%d\n" % (e, f), NOP, MOV, B]
        elif which == "unpackB_late_mov":
            bm = re.match(r"^\tv_pk_add_f32 v\[(\d+):(\d+)\], v\[%d:%d\], v\[%d:%d\] op_sel_hi:\[1,0\] neg_lo:\[0,1\] neg_hi:\[0,1\]" % (a, b, e, f), B)
            assert bm, B
            g, h = int(bm.group(1)), int(bm.group(2))
            out += [A, NOP, "\tv_add_f32_e64 v%d, v%d, -v%d\n" % (g, a, e), "\tv_add_f32_e64 v%d, v%d, -v%d\n" % (h, f, e), MOV]
        else:
            raise SystemExit("unknown ctx variant " + which)
        i += 4
    return out, n


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.exit("FAILED: %s\n%s" % (" ".join(cmd), r.stdout[-3000:]))
    return r.stdout


def main():
    os.makedirs(OUT, exist_ok=True)
    work = os.path.join("/tmp", "hz_build_" + STEM)
    os.makedirs(work, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=fast", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]
    run(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CSRC, SRC), "-o", "packed.o", "-save-temps"], work)
    dev_s = os.path.join(work, STEM + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    host_s = os.path.join(work, STEM + "-host-x86_64-unknown-linux-gnu.s")
    dev_lines = open(dev_s).readlines()
    host = open(host_s).read()
    # the fat binary is an .asciz literal in section .hip_fatbin followed by its .size: replace both by .incbin of the variant's bundle
    m = re.search(r'(\.section\s+\.hip_fatbin.*?\n\s*\.p2align\s+12.*?\n)(\.L__unnamed_\d+):\n\t\.asciz\t".*?"\n\t\.size\t\2, \d+\n', host, re.S)
    assert m, "fat binary literal not found in the host assembly"
    variants = [("P0", "none", 0)]
    for spec in (sys.argv[1:] or ["before_pk:4", "after_pk:4", "after_trans:4", "dep_pk:4", "dep_pk:1", "dep_pk:2", "dep_any:4", "pad_fma:4"]):
        mode, n = spec.split(":", 1)
        if mode == "ctx":
            variants.append(("ctx_" + n, "ctx", n))
            continue
        if mode == "keep":                              # keep:a:b -- everything unpacked but indices [a, b) of the target kernel
            a, b = (int(t) for t in n.split(":"))
            variants.append(("keep%d_%d" % (a, b), "keep", (a, b)))
        else:
            variants.append((mode + n, mode, int(n)))
    for name, mode, n in variants:
        if mode == "ctx":
            lines, n_ins = ctx(dev_lines, n)
        elif mode == "keep":
            lines, n_ins = unpack(dev_lines, n[0], n[1])
        else:
            lines, n_ins = transform(dev_lines, mode, n)
        vs = os.path.join(work, "v_%s.s" % name)
        open(vs, "w").writelines(lines)
        run([LLVM + "/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950", "-mrelocation-model", "pic",
             "-o", "v_%s.dev.o" % name, vs], work)
        run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-plugin-opt=-amdgpu-internalize-symbols",
             "-plugin-opt=mcpu=gfx950", "-o", "v_%s.out" % name, "v_%s.dev.o" % name], work)
        run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", "-input=v_%s.out" % name, "-output=v_%s.hipfb" % name], work)
        fb = os.path.join(work, "v_%s.hipfb" % name)
        hs = host[:m.start()] + m.group(1) + m.group(2) + ':\n\t.incbin\t"%s"\n\t.size\t%s, %d\n' % (fb, m.group(2), os.path.getsize(fb)) + host[m.end():]
        hp = os.path.join(work, "v_%s.host.s" % name)
        open(hp, "w").write(hs)
        obj = os.path.join(OUT, "%s_%s.o" % (STEM, name))
        run([LLVM + "/clang", "-cc1as", "-triple", "x86_64-unknown-linux-gnu", "-filetype", "obj", "-target-cpu", "x86-64", "-mrelocation-model", "pic",
             "-o", obj, hp], work)
        print("%-16s %s %s -> %s" % (name, n_ins, "(unpacked, kept packed, not unpackable)" if mode == "keep" else "s_nop inserted", os.path.relpath(obj, ROOT)))


if __name__ == "__main__":
    main()
