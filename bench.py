#!/usr/bin/env python
"""Headline benchmark: megapixels/s, forward+backward, NAFNet-64 (enc [1,1,1,28], mid 1, dec [1,1,1,1]),
256x256, batch 32 per GPU, fp32 (BASELINE.json metric; configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + L1 loss + backward of the whole network (every NAFBlock / conv through
libdcpt_hip.so), gradient all-reduce over RCCL when N > 1 (torch DDP buckets, overlapped with
backward), and the fused AdamW update.  Inputs are synthetic (torch.rand, resident in HBM before the
timed region), weights are the keyed deterministic init.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
BATCH, SIZE = 32, 256
# algorithmic work of one 256x256 image, fwd+bwd
FLOP_PER_IMAGE = 378.3e9
BYTES_PER_IMAGE = 3.50e9
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA (= vector) peak
PEAK_HBM_TBS = 8.0

ALOAD = {0: "plain", 1: "ln", 2: "scale", 3: "sg", 4: "gather", 5: "conv3", 6: "lnbf"}
EPI = {0: "plain", 1: "bias", 2: "resid", 3: "sgbwd", 4: "scatter", 5: "scatter_add", 6: "addscaled", 7: "mul", 8: "biasgate", 9: "dotcol", 10: "lnbwd", 11: "resid+ln"}


def prof_class_name(cls: int) -> str:
    if cls >= 1024:   # the chain kernels of the wide levels (chain_bf16.hip)
        return {1024: "chain_fwd_bf16<ffn: LN2+conv4+gate+conv5+residual>", 1025: "chain_fwd_bf16<head: LN1+conv1>",
                1026: "chain_bwd_mid_bf16<LN2'+conv3^T+sca sums>", 1028: "adamw<all parameter tensors of a group>"}.get(cls, f"chain<{cls - 1024}>")
    if cls == 512 + 257:
        return "gemm_tn_bf16_256<grouped weight gradients of a block>"
    if cls == 512 + 256:
        return "gemm_tn_bf16"
    if 256 <= cls < 512:
        return f"gemm_nt_bf16<epi={ {0: 'plain', 1: 'bias', 2: 'resid', 3: 'sgbwd', 4: 'biasgate', 5: 'dotcol', 6: 'lnbwd2', 7: 'scatter', 8: 'scatter_add'}.get(cls - 256, cls - 256)}>"
    if cls >= 512:
        c = cls - 512
        return f"gemm_tn<x={ALOAD.get(c // 8, c // 8)},y={ALOAD.get(c % 8, c % 8)}>"
    return f"gemm_nt<a={ALOAD.get(cls // 16, cls // 16)},epi={EPI.get(cls % 16, cls % 16)}>"


def cpu_baseline(seconds_budget: float = 25.0):
    """The oracle (CPU restatement of the reference path, proven equal to the reference in the build
    container) timed on this box's host cores: forward + L1 + backward at batch 1, 256x256."""
    from dcpt_amd.keyed_init import keyed_input, keyed_state_dict
    from oracle import nafnet_oracle as O

    # 16 threads is the fastest setting for this graph on the GPU box's EPYC (probed: 16 -> 0.48 s/step,
    # 32 -> 1.15, 64 -> 2.1, 128 -> 9.0 at batch 1): the ops are small and oversubscription hurts.
    nthreads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    P = {k: v.requires_grad_(True) for k, v in keyed_state_dict(O.nafnet_param_shapes(**CFG), seed=0).items()}
    x = keyed_input("bench.cpu.x", (1, 3, SIZE, SIZE))
    gt = keyed_input("bench.cpu.gt", (1, 3, SIZE, SIZE))

    def step():
        for p in P.values():
            p.grad = None
        y, _ = O.nafnet_forward(x, P)
        O.l1_loss(y, gt).backward()

    step()  # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < seconds_budget and len(times) < 8):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": round(SIZE * SIZE / 1e6 / med, 5),
        "unit": "megapixels/s",
        "cores": nthreads,
        "kind": "port",
        "sample": f"oracle (PyTorch-CPU fp32 restatement) fwd+L1+bwd, batch 1 x 256x256, median of {len(times)} steps "
                  f"({med:.2f} s/step), {nthreads} threads",
    }


def lib_digest():
    """sha256 over dcpt_amd/csrc + the header + the flags that libdcpt_hip.so was built from (dcpt_amd/build.py)"""
    try:
        with open(os.path.join(ROOT, "dcpt_amd", "lib", "libdcpt_hip.digest")) as fh:
            return fh.read().strip()
    except OSError:
        return None


def live_roofline(lib, run_one_step):
    """The dominant GEMM-class kernel family of a bf16 step, live: one more step with every launch of the library's GEMM / chain kernels bracketed
    by HIP events on its stream (prof.hip), aggregated per family; algorithmic flops / bytes of the launches against the bf16 MFMA / HBM roofs."""
    try:
        lib.dcpt_prof_enable(1)
        run_one_step()
        torch.cuda.synchronize()
        buf = (ctypes.c_double * (8 * 512))()
        n = lib.dcpt_prof_read(buf, 512)
        lib.dcpt_prof_enable(0)
        by = {}
        for i in range(n):   # per kernel family (the shapes of a family differ by level)
            a = by.setdefault(prof_class_name(int(buf[i * 8])), dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            a["launches"] += int(buf[i * 8 + 4]); a["ms"] += buf[i * 8 + 5]; a["flops"] += buf[i * 8 + 6]; a["bytes"] += buf[i * 8 + 7]
        name, a = max(by.items(), key=lambda kv: kv[1]["ms"])
        tf = a["flops"] / a["ms"] / 1e9
        return {"kernel": name, "bound": "hbm" if a["bytes"] / 8e12 > a["flops"] / 2.5e15 else "mfma", "launches_per_step": a["launches"],
                "ms_per_step": round(a["ms"], 3), "achieved_tflops": round(tf, 1), "peak_tflops": 2500.0, "mfma_frac": round(tf / 2500.0, 4),
                "achieved_gbs": round(a["bytes"] / a["ms"] / 1e6, 1), "hbm_frac": round(a["bytes"] / a["ms"] / 1e6 / 8000.0, 4),
                "avg_launch_us": round(1e3 * a["ms"] / max(1, a["launches"]), 1),
                "measured": "live, HIP events around every launch of the family in one extra step (algorithmic flops / bytes of the launches)"}
    except Exception as e:   # noqa: BLE001  (the roofline object is extra: never lose the timing over it)
        return {"error": str(e)[:200]}


def secondary_naf_bf16(dev, steps=6, warmup=3, act="bf16"):
    """The headline network and batch (NAFNet-64 [1,1,1,28], B = 32, 256 x 256, fwd + L1 + bwd + AdamW) with bf16 STORAGE of the feature maps
    (fp32 accumulate, fp32 parameters / optimizer): the one workload north_star's "40 % of the HBM roofline" can apply to (the
    fp32 step is MFMA-bound by 5.5 x; in bf16 the narrow levels are HBM-bound).  Never the headline."""
    from basicsr.archs import build_network
    from dcpt_amd import _lib
    from dcpt_amd.keyed_init import fill_module_
    from dcpt_amd.optim import FusedAdamW

    net = fill_module_(build_network(dict(type="NAFNetBaseline", act_dtype=act, **CFG))).to(dev)
    optm = FusedAdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.9), weight_decay=0.0)
    g = torch.Generator(device=dev).manual_seed(1234)
    lq = torch.rand((BATCH, 3, SIZE, SIZE), generator=g, device=dev)
    gt = torch.rand((BATCH, 3, SIZE, SIZE), generator=g, device=dev)

    def step():
        optm.zero_grad(set_to_none=True)
        (net(lq) - gt).abs().mean().backward()
        optm.step()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    roof = live_roofline(_lib.load(), step)
    nbytes = BATCH * (25 * 30.146e6 + 3 * 40.4e6) * 2   # SURVEY 8d element passes per 256^2 image, 2 B each
    flops = BATCH * FLOP_PER_IMAGE
    out = {"workload": f"NAFNet-64 [1,1,1,28] fwd+L1+bwd+AdamW, B={BATCH}, {SIZE}x{SIZE}, feature maps {act} (fp32 accumulate / parameters / optimizer)",
           "ms_per_step": round(dt * 1e3, 2), "megapixels_per_s": round(BATCH * SIZE * SIZE / 1e6 / dt, 3), "steps": steps, "warmup": warmup,
           "alg_tflops": round(flops / dt / 1e12, 2), "mfma_frac": round(flops / dt / 2.5e15, 4), "alg_gbytes": round(nbytes / 1e9, 2),
           "hbm_frac": round(nbytes / dt / (PEAK_HBM_TBS * 1e12), 4), "bound": "hbm" if nbytes / 8e12 > flops / 2.5e15 else "mfma",
           "roofline": roof}
    del net, optm
    torch.cuda.empty_cache()
    return out


def secondary_dcpt_bf16(dev, steps=5, warmup=3, sizes=(128, 256), act="bf16"):
    """BASELINE.json configs[2] at its own dtype and size, timed by the run that prints the line: one DCPT pre-training step
    (reference ...pretrain_model.py:133-169: encoder on the clean and on the degraded batch, classifier head on the decoder taps,
    one backward, two AdamW updates) with NAFNet-64 + PromptIR_NoImg_DC([64,128,256,512]), 10 classes, B = 32, encoder AND head in
    bf16 storage (fp32 accumulate, fp32 parameters / optimizer), at 128 x 128 and 256 x 256.  Never the headline."""
    from basicsr.models import build_model
    from dcpt_amd.keyed_init import fill_module_

    out = {}
    naf_bytes_bf16 = (25 * 30.146e6 + 3 * 40.4e6) * 2   # SURVEY 8d element passes per 256^2 image, 2 B each
    for S in sizes:
        B = 32
        opt = dict(name="b", model_type="DCPTModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
                   hook_names="decoder", network_g=dict(type="NAFNetBaseline", act_dtype=act, **CFG),
                   network_dc=dict(type="PromptIR_NoImg_DC", feature_dims=[64, 128, 256, 512], num_res_blocks=2, num_classes=10, act_dtype="bf16"),
                   path=dict(), train=dict(pixel_opt=dict(type="L1Loss"), classify_opt=dict(type="CrossEntropyLoss"),
                                           optim_g=dict(type="AdamW", lr=1e-4, fused=True), optim_dc=dict(type="AdamW", lr=1e-4, fused=True)))
        m = build_model(opt)
        fill_module_(m.net_g)
        fill_module_(m.net_dc)
        g = torch.Generator(device=dev).manual_seed(4321)
        m.feed_data({"lq": torch.rand((B, 3, S, S), generator=g, device=dev), "gt": torch.rand((B, 3, S, S), generator=g, device=dev),
                     "dataset_idx": torch.randint(0, 10, (B,), generator=g, device=dev)})
        for _ in range(warmup):
            m.optimize_parameters(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.optimize_parameters(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        from dcpt_amd import _lib

        roof = live_roofline(_lib.load(), lambda: m.optimize_parameters(1))
        sc = B * (S / 256.0) ** 2
        flops = sc * 1.315e12          # 2 x 378.3 GF (encoder, fwd+bwd) + 558.9 GF (head) per 256^2 image
        log = m.get_current_log()
        out[f"dcpt_all_bf16_{S}" if act == "bf16" else f"dcpt_{act}_{S}"] = {
            "workload": f"DCPT step (BASELINE.json configs[2]): NAFNet-64 x2 fwd + PromptIR_NoImg_DC([64,128,256,512]) head + bwd + 2x AdamW, "
                        f"B={B}, {S}x{S}, encoder feature maps {act}, head bf16 (fp32 accumulate / parameters)",
            "ms_per_step": round(dt * 1e3, 2), "megapixels_per_s": round(B * S * S / 1e6 / dt, 3), "steps": steps, "warmup": warmup,
            "alg_tflops": round(flops / dt / 1e12, 2), "mfma_frac": round(flops / dt / 2.5e15, 4),
            "hbm_frac": round(sc * 2 * naf_bytes_bf16 / dt / (PEAK_HBM_TBS * 1e12), 4),
            "hbm_frac_note": "encoder's algorithmic bf16 bytes only (SURVEY 8d has no byte count for the head): a lower bound",
            "l_pix": round(float(log["l_pix"]), 6), "l_classify": round(float(log["l_classify"]), 6),
            "roofline": roof,
        }
        del m
        torch.cuda.empty_cache()
    return out


def ddp_wrap_overhead(unwrapped_ms: float) -> dict:
    """What the DDP wrap itself costs (reference base_model.py:108-115), on a ONE-rank RCCL group: `python bench.py --force-ddp` as its own
    process right after the headline run on the same GPU -- the process group exists before the first kernel, as under torchrun (wrapping
    a network that has already trained in this process measured 13 % slower for reasons that have nothing to do with the wrap:
    tools/ddp_probe.py vs profiles/r4/ddp_probe.txt) -- against the unwrapped ms_per_step of this run."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--force-ddp", "--no-secondary", "--no-cpu-baseline", "--steps", "6", "--warmup", "4"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT).stdout.strip().splitlines()
        line = json.loads([ln for ln in out if ln.startswith("{")][-1])
        ms = float(line["ms_per_step"])
        return {"workload": "the headline step with the network wrapped in DistributedDataParallel (64 MB buckets as gradient views, built-in "
                            "all-reduce hook, the blocks' gradients written straight into the bucket views: dcpt_amd/ddp.py) on a 1-rank RCCL group",
                "command": "python bench.py --force-ddp --no-secondary --no-cpu-baseline --steps 6 --warmup 4",
                "ms_per_step": round(ms, 3), "unwrapped_ms_per_step": round(unwrapped_ms, 3),
                "ddp_wrap_overhead_ms": round(ms - unwrapped_ms, 3), "ddp_wrap_overhead_frac": round(ms / unwrapped_ms - 1.0, 4),
                "rccl_ranks": line["config"].get("rccl_ranks")}
    except Exception as e:  # noqa: BLE001  (a secondary line must never take the headline down)
        return {"error": repr(e)[:300]}


def spawn_ranks(n: int, script: str = None) -> None:
    """`python bench.py --gpus N` without a launcher: re-execute this command (or ``script``: bench_extra.py) under torch.distributed.run,
    one process per GPU."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"--gpus {n} but only {have} device(s) visible on this node: refusing to report a {have}-GPU number as n_gpus={n}")
    from dcpt_amd import build as _build

    _build.build()   # once, before the ranks exist (the ranks only load the finished library)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script or __file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it across processes)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (the metric is quoted at 32)")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DistributedDataParallel even with one rank (path check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path-check-shared-device", action="store_true",
                    help="PATH CHECK ONLY, never a measurement: the ranks share the visible GPU(s) and the collectives go over gloo, so that the "
                         "multi-rank code path can be exercised on a one-GPU box (the line is marked invalid)")
    ap.add_argument("--gemm-precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="fp32: the exact fp32 MFMA kernels (the headline); bf16x3: the wide GEMMs on the bf16 pipe with split operands "
                         "(fp32-class, reported as its own line -- never the headline)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bf16 DCPT steps (configs[2]) timed after the headline region")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--side-stream", type=int, default=1, help="0: keep the weight-gradient GEMMs on the main stream")
    ap.add_argument("--iso-steps", type=int, default=3, help="steps of the serialized per-kernel timing pass (0 = skip)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one GPU")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- build the library once, then one rank per GPU under
        # torch.distributed.run (the same command line the driver uses); rank 0 of the children prints the JSON line
        return spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if torch.cuda.device_count() < world and not args.path_check_shared_device:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible on this node")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    use_ddp = world > 1 or args.force_ddp
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.path_check_shared_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL

    from basicsr.archs import build_network
    from dcpt_amd import _lib
    from dcpt_amd.keyed_init import fill_module_

    lib = _lib.load()
    lib.dcpt_set_side_stream(1 if args.side_stream else 0)
    if args.gemm_precision != "fp32":
        from dcpt_amd import functional as DF

        DF.set_gemm_precision(args.gemm_precision, device=dev)
    net = build_network(dict(type="NAFNetBaseline", **CFG))
    fill_module_(net, seed=0)
    net = net.to(dev)
    model = net
    if use_ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP

        # reference base_model.py:108-115; buckets all-reduce on RCCL's stream while backward continues
        model = DDP(net, device_ids=[local_rank], bucket_cap_mb=64, gradient_as_bucket_view=True)
        from dcpt_amd import ddp as dcpt_ddp

        dcpt_ddp.prepare(model)   # per-bucket divide + the blocks' gradients written straight into the bucket views (no per-parameter copies)
    from dcpt_amd.optim import FusedAdamW   # torch.optim.AdamW's update on the library's multi-tensor kernel (dcpt_adamw_step)

    opt = FusedAdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.9), weight_decay=0.0)

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    lq = torch.rand((args.batch, 3, SIZE, SIZE), generator=g, device=dev)
    gt = torch.rand((args.batch, 3, SIZE, SIZE), generator=g, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(lq)
        loss = (out - gt).abs().mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()

    def barrier():
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    use_prof = (not args.no_prof)

    def read_prof():
        rows = []
        buf = (ctypes.c_double * (8 * 512))()
        n = lib.dcpt_prof_read(buf, 512)
        lib.dcpt_prof_enable(0)
        for i in range(n):
            cls, M, N, K, cnt, ms, fl, by = (buf[i * 8 + j] for j in range(8))
            rows.append(dict(kernel=prof_class_name(int(cls)), M=int(M), N=int(N), K=int(K), launches=int(cnt), ms=ms,
                             flops=fl, bytes=by))
        rows.sort(key=lambda r: -r["ms"])
        return rows

    SAMPLE = 7   # in the timed region every 7th GEMM launch is bracketed (an event pair costs ~3 us of queue time; 7 is
                 # co-prime with the number of GEMM launches per block, so every launch class gets sampled)
    barrier()
    if use_prof:
        lib.dcpt_prof_enable(SAMPLE)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    loss_val = float(loss.detach())
    prof_rows = read_prof() if use_prof else []
    for r in prof_rows:   # sampled -> per-step totals are extrapolated; per-launch figures are exact for the sampled launches
        for k in ("launches", "ms", "flops", "bytes"):
            r[k] = r[k] * SAMPLE
    # With the weight-gradient side stream the GEMMs of two streams overlap, so an event pair around one launch also
    # spans its neighbours' work: per-kernel durations are taken from a short SERIALIZED pass (same steps, side stream
    # off) after the timed region; the as-run figures of the timed region are reported next to them.
    asrun_rows, iso_ms = None, None
    if use_prof and args.side_stream and args.iso_steps > 0:
        asrun_rows = prof_rows
        lib.dcpt_set_side_stream(0)
        step()
        barrier()
        lib.dcpt_prof_enable(1)
        t1 = time.perf_counter()
        for _ in range(args.iso_steps):
            step()
        barrier()
        iso_ms = (time.perf_counter() - t1) / args.iso_steps * 1e3
        prof_rows = read_prof()
        for r in prof_rows:   # normalise to the timed region's step count
            for k in ("launches", "ms", "flops", "bytes"):
                r[k] = r[k] * args.steps / args.iso_steps
        lib.dcpt_set_side_stream(1)
    # SURVEY 8d defines the step as forward + loss + backward (+ all-reduce), optimizer excluded: the same steps without opt.step(),
    # timed right after the timed region (value / ms_per_step keep the optimizer: the conservative figure)
    nfb = max(1, min(args.steps, 10))

    def step_fb():
        opt.zero_grad(set_to_none=True)
        (model(lq) - gt).abs().mean().backward()
        if use_ddp:   # (what the optimizer's post-step hook does in a full step: the bucket views may be written in place again)
            dcpt_ddp.refresh_views(net.parameters())

    step_fb()
    barrier()
    t_fb = time.perf_counter()
    for _ in range(nfb):
        step_fb()
    barrier()
    dt_fb = (time.perf_counter() - t_fb) / nfb
    rccl_ranks = None
    if use_ddp:
        tf = torch.tensor([dt_fb], device=dev, dtype=torch.float64)
        dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        dt_fb = float(tf.item())
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)   # every rank that took part in the timed region contributes 1
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        mp = world * args.batch * SIZE * SIZE / 1e6
        value = mp / (dt / args.steps)
        res = {
            "metric": "megapixels/sec fwd+bwd NAFNet-64 256px bs=32",
            "value": round(value, 3),
            "unit": "megapixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "ms_fwd_bwd": round(dt_fb * 1e3, 3),
            "value_fwd_bwd": round(mp / dt_fb, 3),
            "fwd_bwd_note": f"forward + L1 + backward (+ all-reduce) WITHOUT the optimizer (SURVEY 8d's step), {nfb} steps timed right after the "
                            "timed region; value / ms_per_step include the fused AdamW update",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.gemm_precision == "fp32" else "f32 via bf16x3 split operands",
            "data": "synthetic (torch.rand images, keyed deterministic weights)",
            "config": {
                "workload": "NAFNet-width64 enc[1,1,1,28] mid1 dec[1,1,1,1] fwd+L1+bwd(+all-reduce)+AdamW, 256x256, fp32 "
                            "(BASELINE.json configs[1])",
                "per_gpu_batch": args.batch, "global_batch": world * args.batch, "image": [SIZE, SIZE],
                "parallelism": f"dp{world}", "rccl_ranks": rccl_ranks, "loss": round(loss_val, 6),
                "wgrad_side_stream": bool(args.side_stream),
            },
        }
        step_s = dt / args.steps
        # the roof of the matrix pipe the GEMMs run on: exact fp32 MFMA, or -- in the split-operand mode -- the bf16 pipe at six piece
        # products per fp32 multiply-add
        peak_tf = PEAK_F32_TFLOPS if args.gemm_precision == "fp32" else 2500.0 / 6.0
        whole = {
            "mfma_frac": round(args.batch * FLOP_PER_IMAGE / step_s / (peak_tf * 1e12), 4),
            "hbm_frac": round(args.batch * BYTES_PER_IMAGE / step_s / (PEAK_HBM_TBS * 1e12), 4),
        }
        if prof_rows:
            top = prof_rows[0]
            ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
            # HBM bytes per launch of that kernel from the committed PMC passes (cannot be read live): see the file's "source"
            # (`traffic_stale`: the file was made from a different build of libdcpt_hip.so than the one running now -- digests differ)
            traffic, traffic_src, traffic_stale = None, None, None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                    pmc = json.load(fh)
                if top["kernel"] in pmc["kernels"]:
                    traffic = pmc["kernels"][top["kernel"]]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/pmc_traffic.json (rocprofv3 PMC passes of an earlier run of this command)"
                    traffic_stale = pmc.get("lib_digest") is None or pmc.get("lib_digest") != lib_digest()
            except (OSError, ValueError, KeyError):
                pass
            gemm_ms = sum(r["ms"] for r in prof_rows)
            res["roofline"] = {
                "bound": "mfma", "kernel": top["kernel"], "MNK": [top["M"], top["N"], top["K"]],
                "achieved": round(ach, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s" if args.gemm_precision == "fp32" else "TFLOP/s fp32-equivalent",
                "frac": round(ach / peak_tf, 4),
                "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                "launches": top["launches"], "avg_launch_us": round(top["ms"] * 1e3 / max(1, top["launches"]), 2),
                "alg_flops_per_launch": round(top["flops"] / max(1, top["launches"])),
                "alg_bytes_per_launch": round(top["bytes"] / max(1, top["launches"])),
                "all_gemm_ms_per_step": round(gemm_ms / args.steps, 3),
                "all_gemm_tflops": round(sum(r["flops"] for r in prof_rows) / (gemm_ms * 1e-3) / 1e12, 2),
                "whole_step": whole,
                "measured": ("HIP events around every launch of the kernel, on its stream, " +
                             (f"in a serialized pass of {args.iso_steps} steps right after the timed region (weight-gradient side "
                              f"stream off, {iso_ms:.1f} ms/step); as_run = events around every {SAMPLE}th GEMM launch inside the timed "
                              "region, where launches of the two streams overlap" if asrun_rows is not None else
                              f"around every {SAMPLE}th GEMM launch inside the timed region")),
                "by_kernel": [dict(kernel=r["kernel"], MNK=[r["M"], r["N"], r["K"]], launches=r["launches"],
                                   ms_per_step=round(r["ms"] / args.steps, 3),
                                   tflops=round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                                   alg_gbs=round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)) for r in prof_rows[:16]],
            }
            if asrun_rows:
                same = [r for r in asrun_rows if (r["kernel"], r["M"], r["N"], r["K"]) == (top["kernel"], top["M"], top["N"], top["K"])]
                if same:
                    a = same[0]["flops"] / (same[0]["ms"] * 1e-3) / 1e12
                    res["roofline"]["as_run"] = {"achieved": round(a, 2), "frac": round(a / peak_tf, 4),
                                                 "avg_launch_us": round(same[0]["ms"] * 1e3 / max(1, same[0]["launches"]), 2)}
        else:
            res["roofline"] = {"bound": "mfma", "achieved": round(whole["mfma_frac"] * peak_tf, 2),
                               "peak": round(peak_tf, 1), "unit": "TFLOP/s", "frac": whole["mfma_frac"], "traffic": None,
                               "whole_step": whole}
        # Whole-step account of the DEFAULT command (two streams) from the committed rocprofv3 kernel trace: main-queue time split into
        # MFMA kernels / bandwidth kernels / torch kernels / idle gaps (they sum to the traced step time), side-queue busy time and
        # how much of it coincides with main-queue kernels.  It cannot be produced live (rocprofv3 wraps the process), so it is read
        # from profiles/step_budget.json (tools/step_budget.py; `traced_ms_per_step` says which run it describes).
        try:
            with open(os.path.join(ROOT, "profiles", "step_budget.json")) as fh:
                sb = json.load(fh)
            res["roofline"]["step_budget"] = {
                "traced_ms_per_step": sb["step_ms"],
                "main_stream": {"mfma_kernels_ms": sb["main_mfma_ms"], "hbm_kernels_ms": sb["main_hbm_ms"], "torch_kernels_ms": sb["main_torch_ms"],
                                "idle_ms": sb["main_idle_ms"], "sum_ms": sb["check_sum_ms"]},
                "side_stream": {"busy_ms": sb["side_busy_ms"], "mfma_kernels_ms": sb["side_mfma_ms"], "overlapping_main_kernels_ms": sb["side_overlap_ms"],
                                "in_main_idle_ms": sb["side_in_main_idle_ms"]},
                "launches_per_step": {"main": sb["main_launches"], "side": sb["side_launches"]},
                "source": sb.get("source", "profiles/step_budget.json"),
                "stale": sb.get("lib_digest") is None or sb.get("lib_digest") != lib_digest(),
            }
        except (OSError, ValueError, KeyError):
            pass
        res["lib_digest"] = lib_digest()
        if args.path_check_shared_device:
            res["invalid"] = "path check: the ranks shared a device and the collectives went over gloo -- not a measurement"
        if world == 1 and not args.no_secondary:
            sec = {}
            if args.gemm_precision == "fp32":
                # the SAME network / batch / step with the wide GEMMs in the opt-in split-operand mode (gemm_x3.hip: fp32 operands as
                # three bf16 pieces each on the bf16 matrix pipe, fp32 accumulate -- fp32-class, tests/test_gpu_x3.py); its own line
                from dcpt_amd import functional as DF

                DF.set_gemm_precision("bf16x3", device=dev)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(5):
                    l3 = step()
                torch.cuda.synchronize()
                dt3 = (time.perf_counter() - t2) / 5
                x3_rows = []
                if use_prof:   # the mode's own roofline: its dominant GEMM class, HIP events around every launch (the mode runs single-stream)
                    lib.dcpt_prof_enable(1)
                    for _ in range(2):
                        step()
                    torch.cuda.synchronize()
                    x3_rows = read_prof()
                misses = DF.gemm_x3_scratch_misses()
                DF.set_gemm_precision("fp32")   # (also puts the side stream back to what it was)
                X3_PEAK = 2500.0 / 6.0          # six bf16 piece products per fp32 multiply-add on the 2.5 PFLOP/s bf16 pipe
                line = {
                    "workload": "the headline workload (NAFNet-64 fwd+L1+bwd+AdamW, B=32, 256x256) with the level-1..4 NT GEMMs as split-operand "
                                "bf16x3 products (fp32-class results, not the reference's fp32 arithmetic): reported separately, never the headline",
                    "dtype": "f32 via bf16x3 split operands", "ms_per_step": round(dt3 * 1e3, 2),
                    "megapixels_per_s": round(args.batch * SIZE * SIZE / 1e6 / dt3, 3), "steps": 5, "warmup": 3,
                    "loss": round(float(l3.detach()), 6), "wgrad_side_stream": False, "scratch_misses": int(misses),
                    "roofline": {"bound": "mfma", "pipe": "bf16 MFMA, six piece products per fp32 multiply-add", "peak": round(X3_PEAK, 1),
                                 "unit": "TFLOP/s fp32-equivalent",
                                 "whole_step_frac": round(args.batch * FLOP_PER_IMAGE / dt3 / (X3_PEAK * 1e12), 4)},
                }
                if x3_rows:
                    top3 = x3_rows[0]
                    ach3 = top3["flops"] / (top3["ms"] * 1e-3) / 1e12
                    gms = sum(r["ms"] for r in x3_rows)
                    line["roofline"].update({
                        "kernel": top3["kernel"], "MNK": [top3["M"], top3["N"], top3["K"]], "achieved": round(ach3, 2),
                        "frac": round(ach3 / X3_PEAK, 4), "avg_launch_us": round(top3["ms"] * 1e3 / max(1, top3["launches"]), 2),
                        "all_gemm_tflops": round(sum(r["flops"] for r in x3_rows) / (gms * 1e-3) / 1e12, 2),
                        "note": "classes with N < 256 (levels 0-1, the layers between the groups) still run the exact fp32 MFMA kernels inside this step",
                    })
                sec["naf_f32_via_bf16x3"] = line
            del net, model, opt, lq, gt, loss
            torch.cuda.empty_cache()
            sec["naf_bf16_256"] = secondary_naf_bf16(dev)
            sec.update(secondary_dcpt_bf16(dev))
            # the bf16 mode that holds EVERY image inside north_star's 0.01 dB (act_dtype bf16_edge32: the full-resolution level in fp32;
            # tests/test_gpu_configs.py::test_psnr_bf16_storage_vs_fp32) next to plain bf16: what that tolerance costs
            sec.update(secondary_dcpt_bf16(dev, sizes=(256,), act="bf16_edge32"))
            if args.gemm_precision == "fp32":
                sec["ddp_wrap_one_rank"] = ddp_wrap_overhead(ms_per_step)
            # BASELINE.json configs[3] and configs[4] on the same record (bench_extra.py holds the workloads)
            import bench_extra as BX

            sec["restormer_b64_128"] = BX.run_restormer(dev, "balanced", steps=3, warmup=3)   # (the default mode: explicit, independent of free memory)
            torch.cuda.empty_cache()
            sec["infer2k_fp32"] = BX.run_infer2k(dev, "fp32", steps=3, warmup=2)
            sec["infer2k_bf16"] = BX.run_infer2k(dev, "bf16", steps=3, warmup=2)
            sec["infer2k_bf16_edge32"] = BX.run_infer2k(dev, "bf16_edge32", steps=3, warmup=2)
            torch.cuda.empty_cache()
            res["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        try:   # RCCL prints its version banner through C stdio, which is block-buffered on a pipe: push it out BEFORE the line, so that
            ctypes.CDLL(None).fflush(None)   # the JSON line is the last line of the output
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)
    if use_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
