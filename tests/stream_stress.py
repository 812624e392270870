"""Bit-stability of every kernel family next to concurrently running bf16 MFMA GEMMs of ANOTHER stream (GPU box only):

    python tests/stream_stress.py [--reps 12]

For each victim (a layer's forward under no_grad, or its forward + backward) the quiet result is taken first; then the victim runs
``reps`` times on one stream while a second stream runs bf16 NAFBlocks back to back, and every result must equal the quiet one bit
for bit.  Found with this: the ending conv (conv3x3_b2s_kernel<3>) next to the bf16 GEMM kernels -- packed-fp32 instructions with operand
selection, LABNOTES.md 4h and tools/opsel_repro; the library is built without packed fp32 since."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FULL = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])


def build(act):
    from basicsr.archs import build_network
    from dcpt_amd.keyed_init import fill_module_

    return fill_module_(build_network(dict(type="NAFNetBaseline", act_dtype=act, **FULL))).cuda()


def feat(act, b, c, h, w, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    t = torch.rand((b, c, h, w), generator=g, device="cuda") - 0.5
    if act == "bf16":
        t = t.bfloat16()
    return t.contiguous(memory_format=torch.channels_last)


def victims(net, act, B=2, S=256):
    from dcpt_amd import functional as DF

    img = torch.rand((B, 3, S, S), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out = {}
    out["intro"] = lambda: DF.conv3x3_in(img, net.intro.weight, net.intro.bias, out_bf16=act == "bf16")
    f0 = feat(act, B, 64, S, S, 2)
    out["ending"] = lambda: DF.conv3x3_out(f0, net.ending.weight, net.ending.bias, img)
    levels = [(64, S), (128, S // 2), (256, S // 4), (512, S // 8), (1024, S // 16)]
    blocks = [net.encoders[0][0], net.encoders[1][0], net.encoders[2][0], net.encoders[3][0], net.middle_blks[0]]
    for (c, s), blk in zip(levels, blocks):
        x = feat(act, B * (1 if c < 512 else 4), c, s, s, 3 + c)
        out[f"block C={c} fwd"] = lambda blk=blk, x=x: blk(x)

        def fb(blk=blk, x=x):
            xx = x.detach().clone().requires_grad_(True)
            for p in blk.parameters():
                p.grad = None
            with torch.enable_grad():
                y = blk(xx)
                y.backward(x)
            return [y.detach(), xx.grad] + [p.grad for p in blk.parameters()]

        out[f"block C={c} fwd+bwd"] = fb
        if act == "bf16" and c in (256, 512):
            # the same block on enough pixels for the chain kernels of the wide levels (chain_bf16.hip: LN -> 1 x 1 chains per 128-pixel tile,
            # taken when the tiles fill 3/4 of the chip): 24 576 pixels = 192 tiles
            xb = feat(act, 24576 // (s * s), c, s, s, 70 + c)
            out[f"block C={c} (chain kernels) fwd"] = lambda blk=blk, x=xb: blk(x)

            def fbc(blk=blk, x=xb):
                xx = x.detach().clone().requires_grad_(True)
                for p in blk.parameters():
                    p.grad = None
                with torch.enable_grad():
                    y = blk(xx)
                    y.backward(x)
                return [y.detach(), xx.grad] + [p.grad for p in blk.parameters()]

            out[f"block C={c} (chain kernels) fwd+bwd"] = fbc
    if act == "bf16":
        # the classifier head's BottleneckBlock as one library call (dcpt_bottleneck_*_bf16, round 6): LayerNorm forward / masked LayerNorm backward
        # in the conv GEMMs' epilogues -- C = 64: 128-row kernel + the 512 x 128 tile of the 256-row kernel (>= 192 tiles of 512 pixels);
        # C = 128: the [128][256] epilogue of the 256-row kernel (>= 192 tiles of 256 pixels); C = 32: the 64-column form
        from basicsr.archs.degrad_classify_arch import BottleneckBlock
        from dcpt_amd.keyed_init import fill_module_

        for c, b, s in ((64, 2, 224), (128, 1, 224), (32, 2, 40)):
            blk = fill_module_(BottleneckBlock(c, c, bottleneck_channels=2 * c)).cuda()
            x = feat(act, b, c, s, s, 90 + c)

            def fbn(blk=blk, x=x):
                xx = x.detach().clone().requires_grad_(True)
                for p in blk.parameters():
                    p.grad = None
                with torch.enable_grad():
                    y = blk(xx)
                    y.backward(x)
                return [y.detach(), xx.grad] + [p.grad for p in blk.parameters()]

            out[f"head bottleneck C={c} fwd+bwd"] = fbn
    for i, (c, s) in enumerate(levels[:4]):
        x = feat(act, B, c, s, s, 40 + i)
        out[f"down{i}"] = lambda i=i, x=x: net.downs[i](x)
    for i in range(4):
        c, s = levels[4 - i]
        x, sk = feat(act, B, c, s, s, 50 + i), feat(act, B, c // 2, 2 * s, 2 * s, 60 + i)
        out[f"up{i}"] = lambda i=i, x=x, sk=sk: DF.up_ps(x, net.ups[i][0].weight, sk)
    return out


def same(a, b):
    if isinstance(a, (list, tuple)):
        return all(same(x, y) for x, y in zip(a, b))
    return torch.equal(a, b)


def scan(reps=12, only="", verbose=True):
    """-> [(act, victim, launches that differed)] for the victims that were not bit-stable"""
    nets = {a: build(a) for a in ("bf16", "fp32")}
    d_net = nets["bf16"]
    d_in = [feat("bf16", 4, 1024, 34, 34, 7), feat("bf16", 4, 128, 272, 272, 8), feat("bf16", 4, 64, 544, 544, 9)]

    def disturb(n):
        with torch.no_grad():
            for _ in range(n):
                d_net.middle_blks[0](d_in[0])
                from dcpt_amd import functional as DF
                DF.up_ps(d_in[1], d_net.ups[3][0].weight, d_in[2])

    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    failed = []
    for act, net in nets.items():
        for name, fn in victims(net, act).items():
            if only and only not in f"{act} {name}":
                continue
            grad = "bwd" in name
            with torch.set_grad_enabled(grad):
                quiet = fn()
                torch.cuda.synchronize()
                outs = []
                with torch.cuda.stream(sB):
                    disturb(12)
                with torch.cuda.stream(sA):
                    for _ in range(reps):
                        r = fn()
                        outs.append([t.clone() for t in r] if isinstance(r, (list, tuple)) else r)
                with torch.cuda.stream(sB):
                    disturb(12)
                torch.cuda.synchronize()
            bad = sum(not same(o, quiet) for o in outs)
            if verbose:
                print(f"{act} {name:22s} differing from the quiet result: {bad} / {len(outs)}", flush=True)
            if bad:
                failed.append((act, name, bad))
    return failed


def main():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _variant  # noqa: F401  (DCPT_TOOL_LIB selects an experiments/lib build)

    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    failed = scan(args.reps, args.only)
    print("FAILED:" if failed else "all bit-stable", failed)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
