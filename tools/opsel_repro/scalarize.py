"""Rewrite packed-fp32 instructions of conv3x3_b2s_kernel<3, bf16>'s row loop as scalar ones in the compiler's assembly and assemble code
objects (tools/opsel_repro/build.sh).  usage: scalarize.py <workdir with orig.s> name:lo:hi | name:opsel | name:plain | name:nops ..."""
import os
import re
import subprocess
import sys

W = sys.argv[1]
lines = open(os.path.join(W, "orig.s")).read().split("\n")
KERNEL = "_ZN12_GLOBAL__N_118conv3x3_b2s_kernelILi3EtEEvPKT0_PKfS5_S5_Pfiiiii:"
S = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
E = next(i for i in range(S, len(lines)) if "s_endpgm" in lines[i])
# the row loop = from the first basic block that stores the output to the end of the kernel
LOOP0 = next(i for i in range(S, E) if "buffer_store_dword" in lines[i]) - 4
HERE = os.path.dirname(os.path.abspath(__file__))


def pair(tok):
    m = re.match(r"([vs])\[(\d+):(\d+)\]$", tok)
    if not m:
        raise ValueError(tok)
    return m.group(1), int(m.group(2))


def conv(line):
    t = line.strip()
    m = re.match(r"v_pk_(mul|fma|add)_f32\s+(.*)$", t)
    if not m:
        return None
    kind, rest = m.group(1), m.group(2)
    mods = {mm.group(1): [int(x) for x in mm.group(2).split(",")] for mm in re.finditer(r"(op_sel_hi|op_sel):\[([01,]+)\]", rest)}
    rest = re.sub(r"\s*(op_sel_hi|op_sel):\[[01,]+\]", "", rest).strip()
    ops = [o.strip() for o in rest.split(",")]
    dst, srcs = pair(ops[0]), [pair(o) for o in ops[1:]]
    n = len(srcs)
    sel, selhi = mods.get("op_sel", [0] * n), mods.get("op_sel_hi", [1] * n)
    reg = lambda p, h: f"{p[0]}{p[1] + h}"   # noqa: E731
    lo_src, hi_src = [reg(srcs[i], sel[i]) for i in range(n)], [reg(srcs[i], selhi[i]) for i in range(n)]
    dlo, dhi = reg(dst, 0), reg(dst, 1)
    opn = {"mul": "v_mul_f32_e32", "add": "v_add_f32_e32", "fma": "v_fma_f32"}[kind]
    lo_i, hi_i = f"\t{opn} {dlo}, " + ", ".join(lo_src), f"\t{opn} {dhi}, " + ", ".join(hi_src)
    if dlo in hi_src and dhi in lo_src:
        raise RuntimeError("swap conflict: " + t)
    return [hi_i, lo_i] if dlo in hi_src else [lo_i, hi_i]


def build(name, which, nops=False):
    out, k, done = [], 0, 0
    for i, l in enumerate(lines):
        t = l.strip()
        if LOOP0 <= i <= E and t.startswith("v_pk_") and "f32" in t:
            k += 1
            if which(k, t):
                out.extend(conv(l))
                done += 1
                continue
        out.append(l)
        if nops and LOOP0 <= i < E and t.startswith(("v_", "s_", "buffer_", "global_")) and not t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_nop")):
            out.append("\ts_nop 3")
    src = os.path.join(W, name + ".s")
    open(src, "w").write("\n".join(out))
    llvm = "/opt/rocm/lib/llvm/bin/"
    subprocess.check_call([llvm + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", src[:-2] + ".o"])
    subprocess.check_call([llvm + "ld.lld", "-shared", src[:-2] + ".o", "-o", os.path.join(HERE, name + ".hsaco")])
    print(f"{name}: {k} packed-fp32 instructions in the loop, {done} rewritten")


for spec in sys.argv[2:]:
    parts = spec.split(":")
    if parts[1] == "opsel":
        build(parts[0], lambda k, t: "op_sel" in t)
    elif parts[1] == "plain":
        build(parts[0], lambda k, t: "op_sel" not in t)
    elif parts[1] == "nops":
        build(parts[0], lambda k, t: False, nops=True)
    else:
        build(parts[0], lambda k, t, lo=int(parts[1]), hi=int(parts[2]): lo <= k <= hi)
