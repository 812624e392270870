"""Randomized-shape sweep of the bf16 paths that this round's planner decides (grouped 256 x 256-tile weight gradient: whole images per
block, per-image segments, the scaled-copy fallback, ragged pixel counts; the implicit 3 x 3 loader) -- the bodies of the parity tests
in tests/test_gpu_bf16.py (test infrastructure: they check against oracle/) on shapes drawn from a seeded generator instead of the committed list.  GPU box only:

    python tests/fuzz_shapes.py [--seed 0] [--n 40] [--what block,wgrad,convln]

Prints one line per case and a summary; exit code 1 if any case fails (the failing shapes then belong in the committed test list)."""
import argparse
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _variant  # noqa: E402,F401  (DCPT_TOOL_LIB selects an experiments/lib build)

import torch  # noqa: E402


def block_shapes(rng, n):
    out = []
    while len(out) < n:
        c = rng.choice([128, 256, 256, 512, 512, 512, 1024])
        kind = rng.randrange(6)
        if kind == 0:      # P multiple of 64 and >= 512: whole images / segments inside images
            h, w = rng.choice([(16, 32), (32, 32), (24, 24), (40, 40), (32, 48), (64, 64), (8, 64), (40, 48)])
        elif kind == 1:    # ragged pixel counts
            h, w = rng.randint(9, 45), rng.randint(9, 45)
        elif kind == 2:    # small images: P < 512 (scaled copy) and P < 2C (scale on the activations)
            h, w = rng.choice([(8, 8), (16, 16), (12, 20), (16, 24), (8, 32), (20, 20)])
        elif kind == 3:    # one large image
            h, w = rng.choice([(64, 64), (96, 64), (72, 88), (128, 64)])
        else:
            h, w = 8 * rng.randint(1, 8), 8 * rng.randint(1, 8)
        cap = (6 if c >= 512 else 10) * 1024 * 1024      # elements of one activation: bounds the CPU oracle's time
        bmax = max(1, cap // (c * h * w))
        b = rng.randint(1, min(bmax, 40))
        out.append((b, c, h, w))
    return out


def wgrad_shapes(rng, n):
    out = []
    for _ in range(n):
        N, K = 256 * rng.randint(1, 4), 256 * rng.randint(1, 4)
        if rng.random() < 0.2:
            N, K = rng.choice([64, 128, 192]), rng.choice([64, 128, 320])
        M = rng.choice([rng.randint(1, 600), rng.randint(600, 9000), 64 * rng.randint(1, 300), 256 * rng.randint(1, 64) + rng.randint(0, 3)])
        out.append((M, N, K))
    return out


def convln_shapes(rng, n):
    out = []
    while len(out) < n:
        cin, cout = rng.choice([64, 128, 256, 512]), rng.choice([64, 128, 256, 512])
        ks = rng.choice([1, 3, 3])
        h, w = rng.randint(5, 70), rng.randint(5, 70)
        cap = 5 * 1024 * 1024
        if rng.random() < 0.3:   # round 6: 128 / 256 output channels on enough pixels for the 512 x 128 tile (>= 192 tiles of 512) and for the
            cout = rng.choice([128, 128, 256])   # LayerNorm epilogues of the 256-row kernel (>= 192 tiles of 256), ragged last tiles included
            cin = rng.choice([64, 128, 256])
            h, w = rng.randint(180, 260), rng.randint(180, 260)
            cap = 40 * 1024 * 1024
        bmax = max(1, cap // (max(cin, cout) * h * w))
        b = rng.randint(1, min(bmax, 12))
        out.append((b, cin, cout, h, w, ks, rng.random() < 0.5, rng.random() < 0.5))
    return out


def bneck_shapes(rng, n):
    """the head's BottleneckBlock as one library call (dcpt_bottleneck_*_bf16) against the three-call chain, with the kernel families asserted:
    channel counts on both sides of the LayerNorm-epilogue limits, pixel counts on both sides of the 256-row kernels' tile thresholds"""
    out = []
    while len(out) < n:
        c = rng.choice([8, 16, 24, 40, 64, 64, 96, 128, 128, 192, 256])
        if rng.random() < 0.35 and c in (64, 128):
            h, w, b = rng.randint(150, 240), rng.randint(150, 240), rng.randint(1, 3)
        else:
            h, w = rng.randint(3, 48), rng.randint(3, 48)
            b = rng.randint(1, max(1, min(6, (3 * 1024 * 1024) // (2 * c * h * w))))
        out.append((b, c, h, w, True))
    return out


def block32_shapes(rng, n):
    out = []
    while len(out) < n:
        c = rng.choice([8, 16, 24, 32, 64, 64, 96, 128, 256, 512, 1024])
        h, w = rng.randint(3, 40), rng.randint(3, 40)
        cap = 2 * 1024 * 1024
        b = rng.randint(1, max(1, min(cap // (c * h * w), 12)))
        out.append((b, c, h, w))
    return out


def edge_shapes(rng, n):
    # (Cs <= 3 with Cb = 32 / 64 take the MFMA forms: widths past one 32-pixel tile / 30-column band, heights past one row strip)
    return [(rng.randint(1, 4), rng.choice([1, 2, 3, 4]), rng.choice([8, 16, 32, 64, 64, 128]), rng.randint(1, 72), rng.randint(1, 100)) for _ in range(n)]


def downup_shapes(rng, n):
    return [(rng.randint(1, 4), rng.choice([8, 16, 24, 64, 128, 256]), 2 * rng.randint(1, 16), 2 * rng.randint(1, 16)) for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--what", default="block,wgrad,convln,bneck,fp32")
    args = ap.parse_args()
    import test_gpu_bf16 as T

    dev = torch.device("cuda:0")
    rng = random.Random(args.seed)
    cases = []
    what = args.what.split(",")
    if "block" in what:
        cases += [("block", T.test_nafblock_bf16_oracle, (s,)) for s in block_shapes(rng, args.n)]
    if "wgrad" in what:
        cases += [("wgrad", T.test_conv1x1_wgrad_bf16, s) for s in wgrad_shapes(rng, args.n)]
    if "convln" in what:
        cases += [("convln", T.test_conv_ln_bf16_oracle, s) for s in convln_shapes(rng, max(4, args.n // 2))]
    if "bneck" in what:
        import test_gpu_dchead as TD

        cases += [("bneck", TD.test_bottleneck_node, s) for s in bneck_shapes(rng, max(4, args.n // 2))]
    if "fp32" in what:   # the fp32 path's parity tests (tests/test_gpu_parity.py) on drawn shapes
        import test_gpu_parity as TP

        cases += [("block32", TP.test_nafblock_oracle, s) for s in block32_shapes(rng, args.n)]
        cases += [("edge", TP.test_edge_convs, s) for s in edge_shapes(rng, max(4, args.n // 2))]
        cases += [("downup", TP.test_down_up, s) for s in downup_shapes(rng, max(4, args.n // 2))]
    failed = []
    for name, fn, a in cases:
        t0 = time.time()
        try:
            fn(dev, *a)
            status = "ok"
        except Exception as e:  # noqa: BLE001  (a sweep: report and go on)
            tb = traceback.extract_tb(e.__traceback__)[-1]
            status = f"FAIL {type(e).__name__} at line {tb.lineno}: " + (str(e).splitlines() or [tb.line or ""])[0][:300]
            failed.append((name, a))
            if not isinstance(e, AssertionError):
                traceback.print_exc()
        print(f"{name:7s} {a!s:60s} {time.time() - t0:6.1f}s  {status}", flush=True)
    print(f"{len(cases) - len(failed)} / {len(cases)} cases passed (seed {args.seed})")
    for f in failed:
        print("FAILED", f)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
