#!/usr/bin/env python
"""Secondary workloads of BASELINE.json (configs[2..4]) on ONE MI355X -- evidence for DESIGN.md, not the driver's
contract (that is bench.py).  Prints one JSON line per workload.

    python bench_extra.py --workload dcpt|restormer|infer2k [--steps K] [--warmup W]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
NAF = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=["dcpt", "restormer", "infer2k"])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--size", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from basicsr.archs import build_network
    from dcpt_amd.keyed_init import fill_module_

    g = torch.Generator(device=dev).manual_seed(1234)
    if args.workload == "dcpt":
        # configs[2]: NAFNet-64 encoder + PromptIR_NoImg_DC head, 10 classes, one DCPT step (fp32 here; the reference has no AMP)
        B, S = args.batch or 32, args.size or 128
        from basicsr.models import build_model

        opt = dict(name="b", model_type="DCPTModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
                   hook_names="decoder", network_g=dict(type="NAFNetBaseline", **NAF),
                   network_dc=dict(type="PromptIR_NoImg_DC", feature_dims=[64, 128, 256, 512], num_res_blocks=2, num_classes=10),
                   path=dict(), train=dict(pixel_opt=dict(type="L1Loss"), classify_opt=dict(type="CrossEntropyLoss"),
                                           optim_g=dict(type="AdamW", lr=1e-4, fused=True), optim_dc=dict(type="AdamW", lr=1e-4, fused=True)))
        m = build_model(opt)
        fill_module_(m.net_g)
        fill_module_(m.net_dc)
        data = {"lq": torch.rand((B, 3, S, S), generator=g, device=dev), "gt": torch.rand((B, 3, S, S), generator=g, device=dev),
                "dataset_idx": torch.randint(0, 10, (B,), generator=g, device=dev)}
        m.feed_data(data)
        dt = timed(lambda: m.optimize_parameters(1), args.steps, args.warmup)
        flops = B * (S / 256.0) ** 2 * 1.315e12   # 1.315 TFLOP fwd+bwd per 256^2 image
        res = dict(workload=f"DCPT step: NAFNet-64 x2 fwd + PromptIR_NoImg_DC head + bwd + 2x AdamW, B={B}, {S}x{S}, fp32",
                   ms_per_step=round(dt * 1e3, 2), megapixels_per_s=round(B * S * S / 1e6 / dt, 3),
                   alg_tflops=round(flops / dt / 1e12, 2), mfma_frac=round(flops / dt / 157.3e12, 4), log=m.get_current_log())
    elif args.workload == "restormer":
        # configs[3]: Restormer defaults, 128x128, fwd + L1 + bwd + AdamW
        B, S = args.batch or 64, args.size or 128
        net = fill_module_(build_network(dict(type="Restormer"))).to(dev)
        optm = torch.optim.AdamW(net.parameters(), lr=1e-4, fused=True)
        lq = torch.rand((B, 3, S, S), generator=g, device=dev)
        gt = torch.rand((B, 3, S, S), generator=g, device=dev)

        def step():
            optm.zero_grad(set_to_none=True)
            (net(lq) - gt).abs().mean().backward()
            optm.step()

        dt = timed(step, args.steps, args.warmup)
        flops = B * (S / 128.0) ** 2 * 232e9      # 77.45 GF fwd -> 232 GF fwd+bwd per 128^2 image
        res = dict(workload=f"Restormer (dim 48, [4,6,6,8], BiasFree LN) fwd+L1+bwd+AdamW, B={B}, {S}x{S}, fp32",
                   ms_per_step=round(dt * 1e3, 2), megapixels_per_s=round(B * S * S / 1e6 / dt, 3),
                   alg_tflops=round(flops / dt / 1e12, 2), mfma_frac=round(flops / dt / 157.3e12, 4))
    else:
        # configs[4]: 2K image, SRModel.test_tile with 512 tiles / 16 px context, NAFNet-64 inference
        S = args.size or 2048
        from basicsr.models import build_model

        opt = dict(name="b", model_type="SRModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=False,
                   network_g=dict(type="NAFNetBaseline", window_size=16, **NAF), path=dict(), tile=dict(infer_size=512, tile_pad=16),
                   val=dict(save_img=False))
        m = build_model(opt)
        fill_module_(m.net_g)
        img = torch.rand((1, 3, S, S), generator=g, device=dev)

        def run():
            m.feed_data({"lq": img})
            m.pre_test()
            m.test_tile()
            m.post_test()

        dt = timed(run, args.steps, args.warmup)
        flops = (S / 256.0) ** 2 * 126.11e9 * (544 / 512.0) ** 2
        res = dict(workload=f"NAFNet-64 tiled inference, {S}x{S}, test_tile infer_size 512 / tile_pad 16, fp32",
                   ms_per_image=round(dt * 1e3, 2), megapixels_per_s=round(S * S / 1e6 / dt, 3),
                   alg_tflops=round(flops / dt / 1e12, 2), mfma_frac=round(flops / dt / 157.3e12, 4))
    res["peak_mem_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
