"""Account for one training step of the DEFAULT bench command (two streams) from a rocprofv3 --kernel-trace CSV:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o bench -- \
        python <repo>/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-prof
    python tools/step_budget.py <dir>/**/bench_kernel_trace.csv [--json out.json]

A step period is delimited by consecutive launches of a kernel that runs once per step (the ending convolution's forward).  For every complete step:
main queue = the queue that carries that kernel; its busy time is split into MFMA kernels (gemm_nt / gemm_tn), hand-written
bandwidth kernels, and torch kernels (loss, optimizer, copies); `idle` is the part of the step in which the main queue runs
nothing.  main.mfma + main.hbm + main.torch + main.idle == step time by construction.  Side queue(s) = everything else
(the weight-gradient side stream): busy time, and how much of it coincides with main-queue kernels (`overlap`) or fills
main-queue gaps (`in_main_idle`)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


def klass(name):
    if name.startswith("gemm_nt_kernel") or name.startswith("gemm_tn_kernel"):
        return "mfma"
    if name.startswith("at::") or "elementwise" in name or "multi_tensor" in name or "rocclr" in name or "reduce_kernel" in name \
            or name.startswith("Cijk") or "vectorized" in name or "FusedAdam" in name or "fused_adam" in name:
        return "torch"
    return "hbm"


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    path = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Queue_Id"), r.get("Stream_Id")), short(r["Kernel_Name"])))
    rows.sort()
    # a step boundary = a hand-written kernel that is launched exactly once per step (the ending convolution's forward pass);
    # picked as the rarest hand-written kernel name in the trace
    cnt = defaultdict(int)
    for r in rows:
        if klass(r[3]) == "hbm":
            cnt[r[3]] += 1
    rare = min(sorted(cnt), key=lambda k: (cnt[k] if cnt[k] >= 3 else 1 << 30))
    marks = [r for r in rows if r[3] == rare]
    main_q = marks[0][2]
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        steps.append((a[0], b[0]))
    steps = steps[1:] if len(steps) > 2 else steps   # drop the first complete step after the profiler attached
    res = []
    per_kernel = defaultdict(lambda: [0, 0.0])
    for t0, t1 in steps:
        inside = [r for r in rows if r[0] >= t0 and r[0] < t1]
        mains = [r for r in inside if r[2] == main_q]
        sides = [r for r in inside if r[2] != main_q]
        by = defaultdict(list)
        for s, e, _, n in mains:
            by[klass(n)].append((s, min(e, t1)))
        mu = union([(s, min(e, t1)) for s, e, _, _ in mains])
        su = union([(s, min(e, t1)) for s, e, _, _ in sides])
        step = (t1 - t0) / 1e6
        busy = {k: length(union(v)) / 1e6 for k, v in by.items()}
        # kernels of one queue do not overlap each other, so the class unions are disjoint
        idle = step - length(mu) / 1e6
        gaps = intersect([[t0, t1]], [[a[1], b[0]] for a, b in zip([[t0, t0]] + mu, mu + [[t1, t1]]) if b[0] > a[1]])
        side_busy = length(su) / 1e6
        res.append(dict(step_ms=step, main_mfma_ms=busy.get("mfma", 0.0), main_hbm_ms=busy.get("hbm", 0.0), main_torch_ms=busy.get("torch", 0.0),
                        main_idle_ms=idle, side_busy_ms=side_busy, side_overlap_ms=length(intersect(mu, su)) / 1e6,
                        side_in_main_idle_ms=length(intersect(gaps, su)) / 1e6,
                        side_mfma_ms=length(union([(s, e) for s, e, _, n in sides if klass(n) == "mfma"])) / 1e6,
                        main_launches=len(mains), side_launches=len(sides)))
        for s, e, q, n in inside:
            key = ("main " if q == main_q else "side ") + n[:70]
            per_kernel[key][0] += 1
            per_kernel[key][1] += (e - s) / 1e6
    n = len(res)
    avg = {k: round(sum(r[k] for r in res) / n, 3) for k in res[0]}
    avg["steps_averaged"] = n
    avg["check_sum_ms"] = round(avg["main_mfma_ms"] + avg["main_hbm_ms"] + avg["main_torch_ms"] + avg["main_idle_ms"], 3)
    top = sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:24]
    avg["top_kernels"] = [dict(kernel=k, launches_per_step=round(v[0] / n, 1), ms_per_step=round(v[1] / n, 3)) for k, v in top]
    print(json.dumps({k: v for k, v in avg.items() if k != "top_kernels"}, indent=1))
    for t in avg["top_kernels"]:
        print(f"  {t['ms_per_step']:8.3f} ms  x{t['launches_per_step']:6.1f}  {t['kernel']}")
    if out_json:
        avg["source"] = "rocprofv3 --kernel-trace of `python bench.py --steps K --warmup W --no-cpu-baseline --no-secondary --no-prof` (default two-stream configuration); tools/step_budget.py"
        import os
        try:   # the build that was traced (bench.py marks the account stale when the running library differs)
            avg["lib_digest"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dcpt_amd", "lib", "libdcpt_hip.digest")).read().strip()
        except OSError:
            avg["lib_digest"] = None
        json.dump(avg, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
