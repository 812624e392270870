"""Fused narrow-level kernels against the kernel chains they replace, on odd sizes (tuning build: tools/build_variant.sh ffnab
"nafblock.hip nafblock_bf16.hip"; run once per setting with DCPT_TOOL_LIB=<lib> DCPT_FFN_FUSED=<0|1> DCPT_FFN_FUSED_F32=<0|1>, then
`compare a.pt b.pt`).   python tools/ffn_fused_vs_chain.py run out.pt | compare a.pt b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(1, 1, 1), (1, 3, 5), (1, 5, 7), (2, 4, 4), (1, 8, 16), (3, 11, 13), (1, 32, 33), (4, 17, 31), (2, 64, 65), (7, 40, 24), (1, 129, 127)]


def run(path):
    from basicsr.archs.nafnet_arch import NAFBlock
    from dcpt_amd.keyed_init import fill_module_
    from dcpt_amd import functional as DF
    dev = torch.device("cuda:0")
    out = {}
    blk = fill_module_(NAFBlock(64), seed=5).to(dev)
    for (B, H, W) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(B * 1000 + H * 10 + W)
        x = (torch.rand((B, 64, H, W), generator=g, device=dev) * 2 - 1).contiguous(memory_format=torch.channels_last)
        go = (torch.rand((B, 64, H, W), generator=g, device=dev) * 2 - 1).contiguous(memory_format=torch.channels_last)
        for dt in ("fp32", "bf16"):
            blk.zero_grad(set_to_none=True)
            xi = (x.bfloat16() if dt == "bf16" else x).clone().requires_grad_(True)
            y = DF.nafblock_bf16(xi, blk.fused_params()) if dt == "bf16" else blk(xi)
            y.backward(go.bfloat16() if dt == "bf16" else go)
            torch.cuda.synchronize()
            out[(B, H, W, dt)] = dict(y=y.detach().float().cpu(), dx=xi.grad.float().cpu(), **{n: p.grad.float().cpu() for n, p in blk.named_parameters()})
    torch.save(out, path)


def compare(a, b):
    A, B_ = torch.load(a), torch.load(b)
    worst = {}
    for k in A:
        for n in A[k]:
            x, y = A[k][n].double(), B_[k][n].double()
            e = float((x - y).abs().max() / (y.abs().max() + 1e-30))
            dt = k[3]
            if e > worst.get((dt, n), (0, None))[0]:
                worst[(dt, n)] = (e, k)
    for dt in ("fp32", "bf16"):
        w = {n: v for (d, n), v in worst.items() if d == dt}
        top = sorted(w.items(), key=lambda t: -t[1][0])[:4]
        print(dt, "worst scale-relative differences:", [(n, f"{v[0]:.2e}", v[1][:3]) for n, v in top])
    lim = {"fp32": 2e-5, "bf16": 3e-2}
    bad = [(k, v) for k, v in worst.items() if v[0] > lim[k[0]]]
    print("OK" if not bad else f"DIFFER: {bad}")


if __name__ == "__main__":
    run(sys.argv[2]) if sys.argv[1] == "run" else compare(sys.argv[2], sys.argv[3])
