"""Static look at the library's ISA for the instruction neighbourhood that made the ending conv unstable next to another stream's bf16 GEMMs
(LABNOTES.md 4h): a packed-fp32 VALU operation (v_pk_mul/fma/add_f32) that reads a VGPR written only `d` instructions earlier in the same
basic block by a 32-bit VALU operation or filled by a load that an s_waitcnt just before retired.

    for f in dcpt_amd/csrc/*.hip: hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S f -o /tmp/isa/<f>.s
    python tools/isa_pk_scan.py /tmp/isa/*.s [--max-d 2]

Prints, per kernel, the packed reads at distance <= max-d with the writer's opcode.  A finding is a place to look at, not a defect: every
kernel family but that one is bit-stable in tests/stream_stress.py."""
import re
import sys
from collections import defaultdict

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path, max_d):
    kernel = None
    last_write = {}      # vgpr -> (index, opcode)
    idx = 0
    pending_loads = {}   # vgpr -> opcode of the load that will fill it
    found = defaultdict(list)
    for line in open(path):
        line = line.split(";")[0].rstrip()
        if not line:
            continue
        if re.match(r"^[_A-Za-z][\w$.]*:\s*$", line):
            name = line.strip()[:-1]
            if not name.startswith(".L"):
                kernel = name
            last_write, pending_loads, idx = {}, {}, 0   # new basic block
            continue
        s = line.strip()
        if s.startswith(".") or kernel is None:
            continue
        parts = s.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        idx += 1
        if op.startswith("s_waitcnt"):
            for r, lop in pending_loads.items():
                last_write[r] = (idx, lop + " (retired by s_waitcnt)")
            pending_loads = {}
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            last_write, pending_loads = {}, {}
            continue
        if op.startswith("v_pk_") and op.endswith("f32") and len(ops) >= 2:
            for src in ops[1:]:
                for r in regs(src):
                    if r in last_write:
                        d = idx - last_write[r][0]
                        wop = last_write[r][1]
                        if d <= max_d and not wop.startswith("v_pk_"):
                            found[kernel].append((d, op, f"v{r}", wop))
        if op.startswith(("buffer_load", "global_load", "ds_read", "flat_load")) and ops:
            for r in regs(ops[0]):
                pending_loads[r] = op
        elif op.startswith("v_") and ops and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            for r in regs(ops[0]):
                last_write[r] = (idx, op)
    return found


def main():
    max_d = 2
    args = sys.argv[1:]
    if "--max-d" in args:
        i = args.index("--max-d")
        max_d = int(args[i + 1])
        del args[i:i + 2]
    files = args
    total = 0
    for f in files:
        for k, hits in scan(f, max_d).items():
            total += len(hits)
            by = defaultdict(int)
            for d, op, r, wop in hits:
                by[(d, wop.split()[0])] += 1
            print(f"{f.split('/')[-1]:22s} {k[:90]:90s} {len(hits):4d}  " + ", ".join(f"d={d} after {w}: {n}" for (d, w), n in sorted(by.items())))
    print("packed-fp32 reads at distance <=", max_d, "of a non-packed write:", total)


if __name__ == "__main__":
    main()
