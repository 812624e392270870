#!/usr/bin/env python
"""Secondary workloads of BASELINE.json (configs[2..4]) on ONE MI355X -- evidence for DESIGN.md, not the driver's
contract (that is bench.py).  Prints one JSON line per workload.

    python bench_extra.py --workload dcpt|restormer|infer2k|naf [--dtype fp32|bf16] [--steps K] [--warmup W]

``--dtype bf16`` (dcpt, naf, infer2k): every feature map of the encoder in bf16 storage with fp32 accumulation (act_dtype="bf16";
dcpt: the classifier head too unless --head-dtype fp32); images, parameters and the optimizer stay fp32.  Its lines carry BOTH
rooflines: the bf16 MFMA peak (2.5 PF dense) and the HBM roof with the bf16 algorithmic bytes -- in bf16 the network is HBM-bound
(SURVEY 8d).
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


def _barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def timed(fn, steps, warmup):
    """W untimed steps, then K steps bracketed by barrier + synchronize on both sides; with several ranks the MAX over ranks"""
    import torch.distributed as dist

    for _ in range(warmup):
        fn()
    _barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _barrier()
    dt = (time.perf_counter() - t0) / steps
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def run_restormer(dev, save="balanced", steps=5, warmup=2, B=64, S=128, rank=0, world=1):
    """BASELINE.json configs[3]: Restormer defaults (reference restormer_arch.py:234-422), fwd + L1 + bwd (+ gradient all-reduce: the network
    in DistributedDataParallel as base_model.py:108-115 wraps it, when world > 1) + AdamW, fp32."""
    from basicsr.archs import build_network
    from dcpt_amd import functional as DF
    from dcpt_amd.keyed_init import fill_module_

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    torch.cuda.reset_peak_memory_stats()
    DF.set_restormer_save(save)
    net = fill_module_(build_network(dict(type="Restormer"))).to(dev)
    bare = net
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel

        from dcpt_amd import ddp as dcpt_ddp

        net = dcpt_ddp.prepare(DistributedDataParallel(net, device_ids=[dev.index], bucket_cap_mb=64, gradient_as_bucket_view=True))
    from dcpt_amd.optim import FusedAdamW

    optm = FusedAdamW(bare.parameters(), lr=1e-4)
    lq = torch.rand((B, 3, S, S), generator=g, device=dev)
    gt = torch.rand((B, 3, S, S), generator=g, device=dev)

    def step():
        optm.zero_grad(set_to_none=True)
        (net(lq) - gt).abs().mean().backward()
        optm.step()

    dt = timed(step, steps, warmup)
    flops = B * (S / 128.0) ** 2 * 232e9      # 77.45 GF fwd -> 232 GF fwd+bwd per 128^2 image
    return dict(workload=f"Restormer (dim 48, [4,6,6,8], BiasFree LN) fwd+L1+bwd+AdamW, B={B}, {S}x{S}, fp32, saved tensors: {save} "
                         "(BASELINE.json configs[3])",
                ms_per_step=round(dt * 1e3, 2), megapixels_per_s=round(world * B * S * S / 1e6 / dt, 3), steps=steps, warmup=warmup,
                alg_tflops=round(flops / dt / 1e12, 2), mfma_frac=round(flops / dt / 157.3e12, 4),
                peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))


def run_infer2k(dev, dtype="fp32", steps=5, warmup=2, S=2048, streams=2):
    """BASELINE.json configs[4]: one S x S image through SRModel.test_tile (reference sr_model.py:273-361), 512-pixel tiles with 16 pixels
    of context, NAFNet-64 inference."""
    from basicsr.models import build_model
    from dcpt_amd.keyed_init import fill_module_

    g = torch.Generator(device=dev).manual_seed(1234)
    torch.cuda.reset_peak_memory_stats()
    peak = 2.5e15 if dtype.startswith("bf16") else 157.3e12
    opt = dict(name="b", model_type="SRModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=False,
               network_g=dict(type="NAFNetBaseline", window_size=16, **dict(NAF, act_dtype=dtype)), path=dict(),
               tile=dict(infer_size=512, tile_pad=16, streams=streams), val=dict(save_img=False))
    m = build_model(opt)
    fill_module_(m.net_g)
    img = torch.rand((1, 3, S, S), generator=g, device=dev)

    def run():
        m.feed_data({"lq": img})
        m.pre_test()
        m.test_tile()
        m.post_test()

    dt = timed(run, steps, warmup)
    flops = (S / 256.0) ** 2 * 126.11e9 * (544 / 512.0) ** 2
    return dict(workload=f"NAFNet-64 tiled inference, {S}x{S}, test_tile infer_size 512 / tile_pad 16, feature maps {dtype} "
                         f"(BASELINE.json configs[4]), tile batches on {streams} stream(s)",
                ms_per_image=round(dt * 1e3, 2), megapixels_per_s=round(S * S / 1e6 / dt, 3), steps=steps, warmup=warmup,
                alg_tflops=round(flops / dt / 1e12, 2), mfma_peak_tflops=peak / 1e12, mfma_frac=round(flops / dt / peak, 4),
                peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=["dcpt", "restormer", "infer2k", "naf"])
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "bf16_tail32", "bf16_edge32"])
    ap.add_argument("--head-dtype", default=None, choices=["fp32", "bf16"], help="dcpt: classifier-head activations (default: --dtype)")
    ap.add_argument("--restormer-save", default="balanced", choices=["auto", "lean", "balanced", "full"], help="what the Restormer halves keep for backward")
    ap.add_argument("--optimizer", default="dcpt", choices=["dcpt", "torch"], help="A/B: torch = torch.optim.AdamW(fused=True) instead of dcpt_amd.optim.FusedAdamW")
    ap.add_argument("--two-pass", action="store_true", help="dcpt A/B: train.batched_encoder_passes false (the reference's two encoder passes, B each, instead of one pass over 2B)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tile-streams", type=int, default=2, help="infer2k: HIP streams the tile batches run on (tile.streams)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--side-stream", type=int, default=1, choices=[0, 1], help="0: weight-gradient work on the caller's stream (serialized kernel times)")
    ap.add_argument("--gpus", type=int, default=1, help="dcpt / restormer: data-parallel ranks, one process per GPU over RCCL (same protocol as bench.py: "
                                                        "self-spawns under torch.distributed.run, barrier + max over ranks, rccl_ranks on the line)")
    ap.add_argument("--path-check-shared-device", action="store_true",
                    help="PATH CHECK ONLY, never a measurement: the ranks share the visible GPU(s), collectives over gloo (the line is marked invalid)")
    args = ap.parse_args()
    if args.gpus > 1 and args.workload not in ("dcpt", "restormer"):
        raise SystemExit("--gpus N: data-parallel lines exist for the training workloads dcpt (configs[2]) and restormer (configs[3])")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import bench

        if torch.cuda.device_count() < args.gpus and not args.path_check_shared_device:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible on this node")
        if args.path_check_shared_device:   # (spawn_ranks refuses more ranks than devices: the shared-device check launches its ranks itself)
            import socket
            import subprocess

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]))
        return bench.spawn_ranks(args.gpus, script=os.path.abspath(__file__))
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="gloo") if args.path_check_shared_device else dist.init_process_group(backend="nccl", device_id=dev)
    if not args.side_stream:
        from dcpt_amd import _lib

        _lib.load().dcpt_set_side_stream(0)
    from basicsr.archs import build_network
    from dcpt_amd.keyed_init import fill_module_

    if args.optimizer == "torch":   # A/B of the optimizer step: torch's fused kernel behind the same constructor
        import dcpt_amd.optim as _O

        _O.FusedAdamW = lambda params, lr=1e-3, **kw: torch.optim.AdamW(params, lr, **{**kw, "fused": True})
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    bf = args.dtype.startswith("bf16")
    naf = dict(NAF, act_dtype=args.dtype)
    # SURVEY 8d algorithmic HBM bytes of NAFNet-64 fwd+bwd per 256^2 image: 25 element passes over the blocks' sum c*P = 30.146 M
    # + 3 passes over the 40.4 M elements of the layers between the groups; 4 B per element in fp32, 2 B in bf16 storage (end to end)
    naf_bytes = (25 * 30.146e6 + 3 * 40.4e6) * (2 if bf else 4)
    peak = 2.5e15 if bf else 157.3e12

    def rooflines(flops, nbytes, dt):
        return dict(alg_tflops=round(flops / dt / 1e12, 2), mfma_peak_tflops=peak / 1e12, mfma_frac=round(flops / dt / peak, 4),
                    alg_gbytes=round(nbytes / 1e9, 2), hbm_frac=round(nbytes / dt / 8e12, 4),
                    bound="hbm" if nbytes / 8e12 > flops / peak else "mfma",
                    note="hbm_frac against the 8 TB/s spec roof; a streaming kernel reaches ~6.3 TB/s (0.79) on this part")

    if args.workload == "naf":
        # configs[1]'s network and batch, as a bf16-vs-fp32 comparison line (the fp32 headline is bench.py's, never this one)
        B, S = args.batch or 32, args.size or 256
        net = fill_module_(build_network(dict(type="NAFNetBaseline", **naf))).to(dev)
        from dcpt_amd.optim import FusedAdamW

        optm = FusedAdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.9), weight_decay=0.0)
        lq = torch.rand((B, 3, S, S), generator=g, device=dev)
        gt = torch.rand((B, 3, S, S), generator=g, device=dev)

        def step():
            optm.zero_grad(set_to_none=True)
            (net(lq) - gt).abs().mean().backward()
            optm.step()

        dt = timed(step, args.steps, args.warmup)
        sc = B * (S / 256.0) ** 2
        res = dict(workload=f"NAFNet-64 [1,1,1,28] fwd+L1+bwd+AdamW, B={B}, {S}x{S}, feature maps {args.dtype}",
                   ms_per_step=round(dt * 1e3, 2), megapixels_per_s=round(B * S * S / 1e6 / dt, 3), **rooflines(sc * 378.3e9, sc * naf_bytes, dt))
    elif args.workload == "dcpt":
        # configs[2]: NAFNet-64 encoder + PromptIR_NoImg_DC head, 10 classes, one DCPT step (fp32 here; the reference has no AMP)
        B, S = args.batch or 32, args.size or 128
        from basicsr.models import build_model

        opt = dict(name="b", model_type="DCPTModel", scale=1, num_gpu=1, dist=world > 1, rank=rank, world_size=world, is_train=True,
                   hook_names="decoder", network_g=dict(type="NAFNetBaseline", **naf),
                   network_dc=dict(type="PromptIR_NoImg_DC", feature_dims=[64, 128, 256, 512], num_res_blocks=2, num_classes=10,
                                   act_dtype=args.head_dtype or ("bf16" if bf else "fp32")),
                   path=dict(), train=dict(pixel_opt=dict(type="L1Loss"), classify_opt=dict(type="CrossEntropyLoss"),
                                           batched_encoder_passes=not args.two_pass,
                                           optim_g=dict(type="AdamW", lr=1e-4, fused=True), optim_dc=dict(type="AdamW", lr=1e-4, fused=True)))
        m = build_model(opt)   # (world > 1: both networks wrapped in DistributedDataParallel by BaseModel.model_to_device, reference base_model.py:108-115)
        fill_module_(m.get_bare_model(m.net_g))
        fill_module_(m.get_bare_model(m.net_dc))
        data = {"lq": torch.rand((B, 3, S, S), generator=g, device=dev), "gt": torch.rand((B, 3, S, S), generator=g, device=dev),
                "dataset_idx": torch.randint(0, 10, (B,), generator=g, device=dev)}
        m.feed_data(data)
        dt = timed(lambda: m.optimize_parameters(1), args.steps, args.warmup)
        sc = B * (S / 256.0) ** 2
        flops = sc * 1.315e12   # 1.315 TFLOP fwd+bwd per 256^2 image (2 x 378.3 GF encoder + 558.9 GF head)
        res = dict(workload=f"DCPT step: NAFNet-64 x2 fwd + PromptIR_NoImg_DC head + bwd (+ all-reduce of both networks' gradients) + 2x AdamW, "
                            f"B={B} per GPU, {S}x{S}, encoder feature maps {args.dtype}, head {args.head_dtype or args.dtype}",
                   ms_per_step=round(dt * 1e3, 2), megapixels_per_s=round(world * B * S * S / 1e6 / dt, 3), log=m.get_current_log())
        if bf:
            # two rooflines for the mixed step: the encoder's flops on the bf16 pipe + the head's on the fp32 pipe (time bound),
            # and the encoder's bf16 algorithmic bytes (the head's bytes are not in SURVEY 8d and are left out: a lower bound)
            head_peak = 2.5e15 if (args.head_dtype or "bf16") == "bf16" else 157.3e12
            t_mfma = sc * (2 * 378.3e9 / 2.5e15 + 558.9e9 / head_peak)
            t_hbm = sc * 2 * naf_bytes / 8e12
            res.update(alg_tflops=round(flops / dt / 1e12, 2), mfma_time_bound_ms=round(t_mfma * 1e3, 2), mfma_frac=round(t_mfma / dt, 4),
                       hbm_time_bound_ms_encoder_only=round(t_hbm * 1e3, 2), hbm_frac=round(t_hbm / dt, 4),
                       note="fractions = time bound / measured step; encoder bf16 (2.5 PF, bf16 bytes), head on its own pipe's peak")
        else:
            res.update(alg_tflops=round(flops / dt / 1e12, 2), mfma_frac=round(flops / dt / 157.3e12, 4))
    elif args.workload == "restormer":
        res = run_restormer(dev, args.restormer_save, args.steps, args.warmup, args.batch or 64, args.size or 128, rank=rank, world=world)
    else:
        res = run_infer2k(dev, args.dtype, args.steps, args.warmup, args.size or 2048, args.tile_streams)
    res["peak_mem_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    if world > 1:
        import torch.distributed as dist

        ones = torch.ones(1, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ones)   # every rank that took part in the timed region contributes 1
        res.update(n_gpus=world, scaling="weak", parallelism=f"dp{world}", rccl_ranks=int(ones.item()) if dist.get_backend() == "nccl" else None)
        if args.path_check_shared_device:
            res["invalid"] = "path check: the ranks shared a device and the collectives went over gloo -- not a measurement"
        dist.barrier()
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
