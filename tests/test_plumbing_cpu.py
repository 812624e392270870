"""CPU: host-side logic of the drop-in surface -- registries, options, state-dict contract, SRModel
padding / tiling / validation arithmetic, metrics, the test.py CLI, and the 2-rank (gloo) DDP step.
A tiny pure-torch arch is registered HERE (tests only) to exercise the callers without a GPU; the
product archs themselves refuse CPU tensors."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from basicsr.utils.registry import ARCH_REGISTRY, Registry  # noqa: E402


@ARCH_REGISTRY.register()
class _TestConvArch(nn.Module):
    """3x3 conv + residual; ``hook`` kwarg like the reference archs."""

    def __init__(self, img_channel=3, window_size=8, width=4):
        super().__init__()
        self.body = nn.Conv2d(img_channel, img_channel, 3, padding=1)
        with torch.no_grad():
            g = torch.Generator().manual_seed(0)
            self.body.weight.copy_(torch.randn(self.body.weight.shape, generator=g) * 0.1)
            self.body.bias.fill_(0.01)

    def forward(self, x, hook=False):
        return None if hook else self.body(x) + x


def _opt(**over):
    opt = dict(name="t", model_type="SRModel", scale=1, num_gpu=0, dist=False, rank=0, world_size=1, is_train=False,
               network_g=dict(type="_TestConvArch", window_size=16), path=dict(pretrain_network_g=None),
               val=dict(save_img=False, metrics=dict(psnr=dict(type="calculate_psnr", crop_border=0, test_y_channel=False),
                                                     ssim=dict(type="calculate_ssim", crop_border=0, test_y_channel=False))))
    opt.update(over)
    return opt


def test_registry_contract():
    r = Registry("x")

    @r.register()
    class A:  # noqa: D401
        pass

    class B_basicsr:  # noqa: N801
        pass

    r.register(B_basicsr)
    assert r.get("A") is A and "A" in r and list(r.keys()) == ["A", "B_basicsr"]
    assert r.get("B") is B_basicsr  # falls back to <name>_basicsr
    with pytest.raises(KeyError):
        r.get("nope")
    with pytest.raises(AssertionError):
        r.register(A)
    # the names the reference's option files resolve, auto-imported by file-name suffix
    import basicsr.archs  # noqa: F401
    import basicsr.models  # noqa: F401
    from basicsr.utils.registry import MODEL_REGISTRY

    for name in ("NAFNetBaseline", "NAFNet", "Restormer", "Restormer_origin", "PromptIR_NoImg_DC", "PromptIR_DC"):
        assert name in ARCH_REGISTRY, name
    for name in ("SRModel", "DCPTModel", "DCTModel", "DCModel"):
        assert name in MODEL_REGISTRY, name
    # PromptIR_DC: the reference's 101 state-dict keys (mixing_weights first, then conv_embed.{0,1}.{weight,bias})
    net = ARCH_REGISTRY.get("PromptIR_DC")(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10)
    keys = list(net.state_dict().keys())
    assert len(keys) == 101 and keys[:5] == ["mixing_weights", "conv_embed.0.weight", "conv_embed.0.bias", "conv_embed.1.weight",
                                             "conv_embed.1.bias"]
    assert tuple(net.conv_embed[0].weight.shape) == (8, 3, 7, 7) and net.conv_embed[0].stride == (2, 2)
    with pytest.raises(TypeError):
        ARCH_REGISTRY.get("PromptIR_DC")(feature_dims=[8], downsample=False)   # not a kwarg of this head in the reference


def test_parse_options_and_force_yml(tmp_path):
    from basicsr.utils.options import parse_options

    yml = os.path.join(ROOT, "options", "all_in_one", "test", "test_NAFNet_5d.yml")
    opt, args = parse_options(str(tmp_path), is_train=False,
                              argv=["-opt", yml, "--force_yml", "num_gpu=0", "network_g:width=32", "val:save_img=true"])
    assert opt["dist"] is False and opt["rank"] == 0 and opt["world_size"] == 1 and opt["is_train"] is False
    assert opt["num_gpu"] == 0 and opt["network_g"]["width"] == 32 and opt["val"]["save_img"] is True
    assert opt["network_g"]["enc_blk_nums"] == [1, 1, 1, 28] and opt["network_g"]["window_size"] == 16
    assert opt["path"]["results_root"].endswith(os.path.join("results", "NAFNet_5d"))
    assert opt["datasets"]["test_1"]["phase"] == "test"
    with pytest.raises(KeyError):
        parse_options(str(tmp_path), is_train=False, argv=["-opt", yml, "--force_yml", "brand_new_key=1"])


def test_nafnet_state_dict_contract():
    from basicsr.archs import build_network
    from oracle import nafnet_oracle as O

    cfg = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1], window_size=16)
    net = build_network(dict(type="NAFNetBaseline", **cfg))
    sd = net.state_dict()
    cfg.pop("window_size")
    shapes = O.nafnet_param_shapes(**cfg)
    assert list(sd.keys()) == list(shapes.keys()) and len(sd) == 664
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    assert sd["encoders.3.27.norm1.weight"].shape == (512,) and sd["ups.0.0.weight"].shape == (2048, 1024, 1, 1)
    assert sd["decoder0.0.beta"].shape == (1, 512, 1, 1)
    names = [n for n, _ in net.named_modules() if "decoder" in n and n.count(".") == 0]
    assert names == ["decoder0", "decoder1", "decoder2", "decoder3"]  # hook targets of DCPTModel
    with pytest.raises(TypeError):
        build_network(dict(type="NAFNetBaseline", bogus_kwarg=1))
    from dcpt_amd._lib import DcptHipError

    with pytest.raises(DcptHipError):  # no CPU fallback in the product arch
        net(torch.zeros(1, 3, 16, 16))


def test_pad_crop_and_tile_arithmetic():
    from basicsr.models import build_model

    m = build_model(_opt())
    x = torch.rand(1, 3, 250, 253, generator=torch.Generator().manual_seed(1))
    m.feed_data({"lq": x})
    m.pre_test()
    assert m.lq.shape == (1, 3, 256, 256) and (m.mod_pad_h, m.mod_pad_w) == (6, 3)
    assert torch.equal(m.lq[..., :250, :253], x)
    assert torch.equal(m.lq[..., 250:, :253], x[..., 243:249, :].flip(2))  # reflect (no edge repeat)
    m.test()
    m.post_test()
    assert m.output.shape == x.shape
    full = m.net_g(x)
    # interior identical to the unpadded run; the last row/col see reflected instead of zero padding
    assert torch.allclose(m.output[..., :249, :252], full[..., :249, :252], atol=1e-6)

    mt = build_model(_opt(tile=dict(infer_size=512, tile_pad=16), network_g=dict(type="_TestConvArch")))
    big = torch.rand(1, 3, 600, 700, generator=torch.Generator().manual_seed(2))
    mt.feed_data({"lq": big})
    mt.pre_test()
    mt.test_tile()
    mt.post_test()
    assert torch.allclose(mt.output, mt.net_g(big), atol=1e-6)  # 3x3 receptive field < tile_pad: tiling is exact


def test_metrics_match_independent_implementations():
    from scipy.signal import correlate2d

    from basicsr.metrics import calculate_psnr, calculate_ssim

    rs = np.random.RandomState(0)
    a = rs.randint(0, 256, (40, 37, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rs.randint(-9, 10, a.shape), 0, 255).astype(np.uint8)
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    assert abs(calculate_psnr(a, b, 0, input_order="HWC") - 10 * np.log10(255.0 ** 2 / mse)) < 1e-9
    assert calculate_psnr(a, a, 0, input_order="HWC") == float("inf")
    assert abs(calculate_ssim(a, a, 0, input_order="HWC") - 1.0) < 1e-12
    # the reference's calling convention: float BCHW batches in [0,1], quantised inside the metric
    fa, fb = a.transpose(2, 0, 1)[None] / 255.0, b.transpose(2, 0, 1)[None] / 255.0
    assert abs(calculate_psnr(fa, fb, 0, image_range=255.0) - calculate_psnr(a, b, 0, input_order="HWC")) < 1e-12
    assert abs(calculate_ssim(fa, fb, 0, image_range=255.0) - calculate_ssim(a, b, 0, input_order="HWC")) < 1e-12
    two = calculate_psnr(np.concatenate([fa, fb]), np.concatenate([fb, fb * 0.5]), 0)
    assert abs(two - 0.5 * (calculate_psnr(fa, fb, 0) + calculate_psnr(fb, fb * 0.5, 0))) < 1e-12
    # independent SSIM: 2-D 11x11 sigma-1.5 window, 'valid'
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    win = np.outer(g / g.sum(), g / g.sum())
    vals = []
    for c in range(3):
        x, y = a[..., c].astype(np.float64), b[..., c].astype(np.float64)
        f = lambda z: correlate2d(z, win, mode="valid")  # noqa: E731
        mx, my = f(x), f(y)
        sx, sy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
        c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        vals.append((((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sx + sy + c2))).mean())
    assert abs(calculate_ssim(a, b, 0, input_order="HWC") - np.mean(vals)) < 1e-9
    assert abs(calculate_psnr(a, b, 4, input_order="HWC") - calculate_psnr(a[4:-4, 4:-4], b[4:-4, 4:-4], 0, input_order="HWC")) < 1e-12


def test_cli_end_to_end_with_cpu_test_arch(tmp_path):
    import logging

    from basicsr.test import test_pipeline

    logging.getLogger("basicsr").handlers.clear()
    yml = os.path.join(ROOT, "options", "all_in_one", "test", "test_NAFNet_5d.yml")
    argv = ["-opt", yml, "--force_yml", "num_gpu=0", "network_g:type=_TestConvArch"]
    for k in ("width", "enc_blk_nums", "middle_blk_num", "dec_blk_nums"):
        argv.append(f"network_g:{k}=4" if k == "width" else f"network_g:{k}=~")
    # the test arch ignores the NAFNet kwargs that remain: give it a forgiving ctor
    orig = _TestConvArch.__init__

    def lenient(self, img_channel=3, window_size=8, width=4, **_):
        orig(self, img_channel, window_size, width)

    _TestConvArch.__init__ = lenient
    try:
        res = test_pipeline(str(tmp_path), argv=argv)
    finally:
        _TestConvArch.__init__ = orig
    assert set(res.keys()) == {"Rain100L", "CBSD68"}
    for r in res.values():
        assert 5.0 < r["psnr"] < 60.0 and 0.0 < r["ssim"] <= 1.0


def test_cli_refuses_cpu_for_product_arch(tmp_path):
    import logging

    from basicsr.test import test_pipeline
    from dcpt_amd._lib import DcptHipError

    logging.getLogger("basicsr").handlers.clear()
    yml = os.path.join(ROOT, "options", "all_in_one", "test", "test_NAFNet_5d.yml")
    with pytest.raises(DcptHipError):
        test_pipeline(str(tmp_path), argv=["-opt", yml, "--force_yml", "num_gpu=0", "network_g:width=8",
                                           "network_g:enc_blk_nums=[1,1]", "network_g:dec_blk_nums=[1,1]"])


# ------------------------------------------------------------------------------------------------ training plumbing
def test_schedulers_concat_sampler_match_the_reference(golden_dir):
    """lr sequences of MultiStepRestartLR / CosineAnnealingRestartLR, ConcatDataset's index -> (sample, dataset_idx) map and
    EnlargedSampler's per-rank index streams == the reference's (tests/golden/train_plumbing.npz)"""
    from basicsr.data.concat_dataset import ConcatDataset
    from basicsr.data.data_sampler import EnlargedSampler
    from basicsr.models import lr_scheduler as lrs

    g = np.load(os.path.join(golden_dir, "train_plumbing.npz"))

    def run(make, n):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=2e-4)
        sch = make(opt)
        seq = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            opt.step()
            sch.step()
            seq.append(opt.param_groups[0]["lr"])
        return np.array(seq)

    cases = {"multistep": lambda o: lrs.MultiStepRestartLR(o, milestones=[5, 9, 9, 14], gamma=0.5),
             "multistep_restart": lambda o: lrs.MultiStepRestartLR(o, milestones=[3, 6, 13, 16], gamma=0.5, restarts=[0, 10],
                                                                   restart_weights=[1, 0.5]),
             "cosine": lambda o: lrs.CosineAnnealingRestartLR(o, periods=[8, 6, 6], restart_weights=[1, 0.5, 0.25],
                                                              eta_min=[1e-7, 2e-7, 3e-7]),
             "cosine_single": lambda o: lrs.CosineAnnealingRestartLR(o, periods=[20], restart_weights=[1], eta_min=1e-6)}
    for name, make in cases.items():
        np.testing.assert_allclose(run(make, 20), g[name], rtol=1e-12, atol=1e-18, err_msg=name)

    class Toy(torch.utils.data.Dataset):
        def __init__(self, n, tag):
            self.n, self.tag = n, tag

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return {"v": self.tag * 100 + i}

    cat = ConcatDataset([Toy(3, 1), Toy(5, 2), Toy(2, 3)], [2, 1, 3])
    assert len(cat) == int(g["concat_len"])
    mine = [[cat[i]["v"], cat[i]["dataset_idx"]] for i in range(len(cat))] + [[cat[-1]["v"], cat[-1]["dataset_idx"]]]
    assert np.array_equal(np.array(mine), g["concat_map"])
    with pytest.raises(ValueError):
        cat[-len(cat) - 1]
    for world, rank, ratio in ((1, 0, 1), (2, 1, 3), (4, 2, 2)):
        sp = EnlargedSampler(Toy(11, 0), world, rank, ratio)
        seqs = []
        for ep in (0, 3):
            sp.set_epoch(ep)
            seqs.append(list(iter(sp)))
        assert np.array_equal(np.array(seqs), g[f"sampler_{world}_{rank}_{ratio}"]) and len(sp) == len(seqs[0])


def _train_yaml(tmp_path, total_iter, **extra):
    import yaml

    opt = dict(name="cpu_train", model_type="SRModel", scale=1, num_gpu=0, manual_seed=3,
               datasets=dict(train_1=dict(name="a", type="SyntheticPairedDataset", num=6, size=48, seed=1, gt_size=32, use_hflip=True,
                                          use_rot=True, batch_size_per_gpu=2, num_worker_per_gpu=0, enlarge_ratio=2),
                             train_2=dict(name="b", type="SyntheticPairedDataset", num=4, size=40, seed=2, gt_size=32, sigma_range=50,
                                          enlarge_ratio=1),
                             val_1=dict(name="v", type="SyntheticPairedDataset", num=2, size=32, seed=9)),
               network_g=dict(type="_TestConvArch", window_size=8),
               path=dict(pretrain_network_g=None, resume_state=None),
               train=dict(total_iter=total_iter, warmup_iter=3, optim_g=dict(type="Adam", lr=2e-3),
                          scheduler=dict(type="MultiStepLR", milestones=[6, 10], gamma=0.5),
                          pixel_opt=dict(type="L1Loss", loss_weight=1.0, reduction="mean"), ema_decay=0.9),
               val=dict(val_freq=5, save_img=False, metrics=dict(psnr=dict(type="calculate_psnr", crop_border=0, test_y_channel=False))),
               logger=dict(print_freq=2, save_checkpoint_freq=4))
    opt.update(extra)
    path = tmp_path / "train.yml"
    path.write_text(yaml.safe_dump(opt))
    return str(path)


def test_train_loop_checkpoint_and_resume(tmp_path):
    """basicsr/train.py end to end on CPU with the test arch: concatenated train sets, warm-up + MultiStepLR, EMA, periodic
    validation and checkpoints, then --auto_resume from the newest state continues at the right iteration and learning rate"""
    import logging

    from basicsr.train import train_pipeline

    logging.getLogger("basicsr").handlers.clear()
    model, res = train_pipeline(str(tmp_path), argv=["-opt", _train_yaml(tmp_path, 9)])
    exp = tmp_path / "experiments" / "cpu_train"
    assert (exp / "models" / "net_g_8.pth").exists() and (exp / "models" / "net_g_latest.pth").exists()
    assert (exp / "training_states" / "8.state").exists()
    assert set(torch.load(exp / "models" / "net_g_8.pth").keys()) == {"params", "params_ema"}
    assert abs(model.get_current_learning_rate()[0] - 1e-3) < 1e-12        # 2e-3 halved at iteration 6
    assert 5.0 < res["v"]["psnr"] < 80.0
    first = model.get_current_log()["l_pix"]
    # resume: 8.state -> runs iterations 9..12 of a 12-iteration schedule (second milestone at 10)
    model2, _ = train_pipeline(str(tmp_path), argv=["-opt", _train_yaml(tmp_path, 12), "--auto_resume"])
    assert (exp / "training_states" / "12.state").exists()
    assert abs(model2.get_current_learning_rate()[0] - 5e-4) < 1e-12
    st = torch.load(exp / "training_states" / "12.state", weights_only=False)
    assert st["iter"] == 12 and st["optimizers"][0]["state"][0]["step"] == 12   # Adam continued from step 8, not from 0
    assert model2.get_current_log()["l_pix"] < 1.5 * first
    # the EMA history survives the restart (reference sr_model.py:70-79): a model built on net_g_8.pth starts its EMA network
    # from the checkpoint's params_ema, not from a copy of params
    from basicsr.models import build_model
    from basicsr.utils.options import parse_options

    ck = torch.load(exp / "models" / "net_g_8.pth")
    assert any(not torch.equal(ck["params"][k], ck["params_ema"][k]) for k in ck["params"])
    opt, _ = parse_options(str(tmp_path), is_train=True, argv=["-opt", _train_yaml(tmp_path, 12)])
    opt["path"]["pretrain_network_g"] = str(exp / "models" / "net_g_8.pth")
    m3 = build_model(opt)
    for k, v in m3.net_g_ema.state_dict().items():
        assert torch.equal(v, ck["params_ema"][k]), k
    for k, v in m3.net_g.state_dict().items():
        assert torch.equal(v, ck["params"][k]), k


# ------------------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import tests.test_plumbing_cpu  # noqa: F401  (registers _TestConvArch in the spawned process)
    from basicsr.models import build_model

    opt = _opt(dist=True, rank=rank, world_size=world, is_train=True,
               train=dict(pixel_opt=dict(type="L1Loss", loss_weight=1.0, reduction="mean"),
                          optim_g=dict(type="SGD", lr=0.0)))
    m = build_model(opt)
    g = torch.Generator().manual_seed(100 + rank)
    m.feed_data({"lq": torch.rand(2, 3, 16, 16, generator=g), "gt": torch.rand(2, 3, 16, 16, generator=g)})
    m.optimize_parameters(1)
    grads = torch.cat([p.grad.flatten() for p in m.net_g.parameters()])
    torch.save({"grads": grads, "log": dict(m.get_current_log())}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_ddp_step_gloo(tmp_path):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(2))
    assert torch.allclose(r0["grads"], r1["grads"], atol=1e-7)  # DDP averaged the gradients
    # single-process reference: mean of the two ranks' local gradients
    from basicsr.models import build_model

    local = []
    losses = []
    for rank in range(2):
        m = build_model(_opt(is_train=True, train=dict(pixel_opt=dict(type="L1Loss"), optim_g=dict(type="SGD", lr=0.0))))
        g = torch.Generator().manual_seed(100 + rank)
        m.feed_data({"lq": torch.rand(2, 3, 16, 16, generator=g), "gt": torch.rand(2, 3, 16, 16, generator=g)})
        m.optimize_parameters(1)
        local.append(torch.cat([p.grad.flatten() for p in m.net_g.parameters()]))
        losses.append(m.get_current_log()["l_pix"])
    assert torch.allclose(r0["grads"], (local[0] + local[1]) / 2, atol=1e-6)
    assert abs(r0["log"]["l_pix"] - (losses[0] + losses[1]) / 2) < 1e-6  # reduce(dst=0) / world


def _tile_shard_worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import tests.test_plumbing_cpu  # noqa: F401
    from basicsr.models import build_model

    m = build_model(_opt(dist=True, rank=rank, world_size=world, network_g=dict(type="_TestConvArch"),
                         tile=dict(infer_size=64, tile_pad=8, max_batch=2, shard_across_ranks=True)))
    big = torch.rand(1, 3, 200, 150, generator=torch.Generator().manual_seed(5))

    class _Loader(list):
        dataset = type("D", (), {"opt": {"name": "v"}})()

    loader = _Loader([{"lq": big, "gt": (big * 0.9).clamp(0, 1), "lq_path": ["a.png"]}])
    res = m.dist_validation(loader, 1, None, False)
    torch.save({"res": dict(res) if res else None}, os.path.join(out, f"t{rank}.pt"))
    m.feed_data({"lq": big})
    m.pre_test()
    m.test_tile()
    m.post_test()
    torch.save({"out": m.output}, os.path.join(out, f"o{rank}.pt"))
    dist.destroy_process_group()


def test_tile_sharding_two_ranks_gloo(tmp_path):
    """``tile.shard_across_ranks`` (SURVEY 8e, configs[4]): tile batches go round-robin to the ranks, one reduce(SUM) to rank 0 assembles
    the image; rank 0's result equals the single-process tiled result and scores the metrics."""
    import torch.multiprocessing as mp

    from basicsr.models import build_model

    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_tile_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    big = torch.rand(1, 3, 200, 150, generator=torch.Generator().manual_seed(5))
    m = build_model(_opt(network_g=dict(type="_TestConvArch"), tile=dict(infer_size=64, tile_pad=8, max_batch=2)))
    m.feed_data({"lq": big})
    m.pre_test()
    m.test_tile()
    m.post_test()
    o0 = torch.load(os.path.join(tmp_path, "o0.pt"))["out"]
    o1 = torch.load(os.path.join(tmp_path, "o1.pt"))["out"]
    assert torch.equal(o0, m.output)                       # assembled on rank 0, bit-identical to the one-process loop
    assert not torch.equal(o1, m.output) and float((o1 == 0).float().mean()) > 0.1   # rank 1 only holds its own tiles
    r0 = torch.load(os.path.join(tmp_path, "t0.pt"))["res"]
    assert r0 is not None and 5.0 < r0["psnr"] < 80.0


def test_psnr_matches_reference_fixture():
    """basicsr.metrics.calculate_psnr vs the reference's own function (tests/golden/metrics.npz, oracle/make_golden.py::gen_metrics):
    same interface -- float BCHW / BHWC batches in [0,1], uint8 rounding inside, border crop, BT.601 luma, batch mean."""
    from basicsr.metrics import calculate_psnr
    from dcpt_amd.keyed_init import keyed_input

    g = np.load(os.path.join(ROOT, "tests", "golden", "metrics.npz"))
    a = keyed_input("metrics.a", (2, 3, 24, 20)).numpy()
    b = np.clip(a + keyed_input("metrics.n", (2, 3, 24, 20), lo=-0.08, hi=0.08).numpy(), 0, 1).astype(np.float32)
    for cb in (0, 3):
        for ych in (False, True):
            mine = calculate_psnr(a, b, cb, test_y_channel=ych, image_range=255.0)
            assert abs(mine - float(g[f"psnr_cb{cb}_y{int(ych)}"])) < 1e-9, (cb, ych, mine)
    assert abs(calculate_psnr(a, b, 0, image_range=1) - float(g["psnr_range1"])) < 1e-9
    assert abs(calculate_psnr(a[0], b[0], 2, image_range=255.0) - float(g["psnr_single_chw"])) < 1e-9
    assert abs(calculate_psnr(a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1), 0, input_order="BHWC") - float(g["psnr_bhwc"])) < 1e-9
    assert calculate_psnr(a, a.copy(), 0) == float("inf") and np.isinf(g["psnr_equal"])
    with pytest.raises(ValueError):
        calculate_psnr(a, b, 0, input_order="NCHW")


def test_denoise_dataset_noise_matches_reference(tmp_path):
    """PairedImageDenoiseDataset draws the reference's noise field (paired_image_dataset.py:388-402): seed 0 outside training,
    seed = index in the train phase, normal(0, sigma/255) on the HWC RGB image -- pinned by tests/golden/denoise_noise.npz."""
    from PIL import Image

    from basicsr.data import build_dataset

    g = np.load(os.path.join(ROOT, "tests", "golden", "denoise_noise.npz"))
    root = tmp_path / "gt"
    root.mkdir()
    for i in range(4):   # four identical files: index 3 exists for the train-phase seed
        Image.fromarray(g["img_hwc_u8"]).save(root / f"{i:02d}.png")
    common = dict(name="n", type="PairedImageDenoiseDataset", dataroot_gt=str(root), io_backend=dict(type="disk"),
                  sigma_type="constant", sigma_range=25)
    test_set = build_dataset(dict(common, phase="test"))
    assert len(test_set) == 4
    for idx in (0, 2):   # seed 0 for every test image
        s = test_set[idx]
        assert s["lq"].shape == (3, 6, 5) and s["lq"].dtype == torch.float32
        assert np.array_equal(s["lq"].permute(1, 2, 0).numpy(), g["lq_test"]), idx
        assert np.array_equal((s["gt"].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8), g["img_hwc_u8"])
    train_set = build_dataset(dict(common, phase="train", use_hflip=False, use_rot=False))
    assert np.array_equal(train_set[3]["lq"].permute(1, 2, 0).numpy(), g["lq_train_idx3"])
    assert not np.array_equal(train_set[1]["lq"].numpy(), train_set[3]["lq"].numpy())   # per-index noise
    # sigma_type random / choice draw sigma from Python's `random` (:388-393)
    import random

    random.seed(5)
    want = random.choice([15, 25, 50])
    random.seed(5)
    s = build_dataset(dict(common, phase="test", sigma_type="choice", sigma_range=[15, 25, 50]))[0]
    rs = np.random.RandomState(0)
    ref = (g["img_hwc_u8"].astype(np.float32) / 255.0)
    ref += rs.normal(0, want / 255.0, ref.shape)
    assert np.array_equal(s["lq"].permute(1, 2, 0).numpy(), ref)
    with pytest.raises(FileNotFoundError):
        build_dataset(dict(common, phase="train", dataroot_gt=str(tmp_path / "missing")))


# ------------------------------------------------------------------------------------------------
@ARCH_REGISTRY.register()
class _TestHookEnc(nn.Module):
    """a two-group encoder with ``decoder{i}`` block groups (hook targets ``decoder{i}.0``) and the reference's ``hook`` contract"""

    def __init__(self, img_channel=3, width=4):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.intro = nn.Conv2d(img_channel, width, 3, padding=1)
        self.decoder0 = nn.Sequential(nn.Conv2d(width, width, 3, padding=1))
        self.decoder1 = nn.Sequential(nn.Conv2d(width, width, 1))
        self.ending = nn.Conv2d(width, img_channel, 3, padding=1)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)

    def forward(self, x, hook=False):
        f = self.decoder1(torch.tanh(self.decoder0(self.intro(x))))
        return None if hook else self.ending(f) + x


@ARCH_REGISTRY.register()
class _TestHead(nn.Module):
    """(lq, taps) -> logits like PromptIR_NoImg_DC: ignores lq, mean-pools every tap"""

    def __init__(self, width=4, num_classes=5, n_taps=2):
        super().__init__()
        self.fc = nn.Linear(width * n_taps, num_classes)
        with torch.no_grad():
            g = torch.Generator().manual_seed(4)
            self.fc.weight.copy_(torch.randn(self.fc.weight.shape, generator=g))
            self.fc.bias.zero_()

    def forward(self, lq, feats):
        return self.fc(torch.cat([f.float().mean(dim=(2, 3)) for f in feats], 1))


def _dcpt_opt(**over):
    opt = dict(name="t", model_type="DCPTModel", scale=1, num_gpu=0, dist=False, rank=0, world_size=1, is_train=True, hook_names="decoder",
               network_g=dict(type="_TestHookEnc"), network_dc=dict(type="_TestHead"), path=dict(),
               train=dict(pixel_opt=dict(type="L1Loss", loss_weight=1.0, reduction="mean"), classify_opt=dict(type="CrossEntropyLoss", loss_weight=1.0),
                          optim_g=dict(type="SGD", lr=0.0), optim_dc=dict(type="SGD", lr=0.0)))
    opt.update(over)
    return opt


def _dcpt_data(rank):
    g = torch.Generator().manual_seed(200 + rank)
    return {"lq": torch.rand(2, 3, 12, 12, generator=g), "gt": torch.rand(2, 3, 12, 12, generator=g), "dataset_idx": torch.randint(0, 5, (2,), generator=g)}


def _dcpt_ddp_worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import tests.test_plumbing_cpu  # noqa: F401
    from basicsr.models import build_model

    m = build_model(_dcpt_opt(dist=True, rank=rank, world_size=world))
    assert m.batched_encoder_passes and len(m.hooks) == 2
    m.feed_data(_dcpt_data(rank))
    m.optimize_parameters(1)
    grads = torch.cat([p.grad.flatten() for net in (m.net_g, m.net_dc) for p in net.parameters()])
    torch.save({"grads": grads, "log": dict(m.get_current_log())}, os.path.join(out, f"d{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_dcpt_step_batched_encoder_pass_gloo(tmp_path):
    """The default DCPT step (``train.batched_encoder_passes``: ONE stacked encoder forward serves the reconstruction and the taps)
    under DistributedDataParallel with world_size 2: both ranks end with the mean of the two ranks' gradients of the REFERENCE's
    two-pass step (...pretrain_model.py:133-169), the hooks fire once per rank and step, losses are averaged on rank 0."""
    import torch.multiprocessing as mp

    from basicsr.models import build_model

    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_dcpt_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"d{i}.pt")) for i in range(2))
    assert torch.allclose(r0["grads"], r1["grads"], atol=1e-7)
    local, logs = [], []
    for rank in range(2):
        opt = _dcpt_opt()
        opt["train"]["batched_encoder_passes"] = False   # the reference's literal two encoder passes, one process
        m = build_model(opt)
        assert not m.batched_encoder_passes
        m.feed_data(_dcpt_data(rank))
        m.optimize_parameters(1)
        local.append(torch.cat([p.grad.flatten() for net in (m.net_g, m.net_dc) for p in net.parameters()]))
        logs.append(m.get_current_log())
    assert float(local[0].abs().max()) > 1e-3
    assert torch.allclose(r0["grads"], (local[0] + local[1]) / 2, atol=2e-6)
    for k in ("l_pix", "l_classify"):
        assert abs(r0["log"][k] - (logs[0][k] + logs[1][k]) / 2) < 1e-6


def test_batched_encoder_pass_detects_split_forward():
    """A net_g that fires its hooks more than once per forward (e.g. one that splits the batch internally) must not silently feed the
    head mis-sliced taps."""
    from basicsr.models import build_model

    m = build_model(_dcpt_opt())
    inner = m.net_g.forward
    m.net_g.forward = lambda x, hook=False: torch.cat([inner(c, hook) for c in x.chunk(2, 0)], 0)
    m.feed_data(_dcpt_data(0))
    with pytest.raises(RuntimeError, match="batched encoder pass"):
        m.optimize_parameters(1)


def test_selfensemble_matches_reference_construction():
    """SRModel.test_selfensemble (reference sr_model.py:187-232) == the reference's own list construction done with numpy: 8
    augmentations (v, h, t applied cumulatively), network on each, inverse transforms in the reference's order, mean -- on a
    non-square image (transposed augmentations have the other shape) -- and ``ensemble: true`` routes validation through it."""
    from basicsr.models import build_model

    m = build_model(_opt(network_g=dict(type="_TestConvArch"), ensemble=True))
    lq = torch.rand(2, 3, 10, 14, generator=torch.Generator().manual_seed(9))
    m.feed_data({"lq": lq})
    m.test_selfensemble()
    assert m.output.shape == lq.shape

    def tf(v, op):   # the reference's _transform, on numpy
        a = v.numpy()
        a = a[..., ::-1] if op == "v" else a[..., ::-1, :] if op == "h" else a.transpose((0, 1, 3, 2))
        return torch.from_numpy(a.copy())

    lst = [lq]
    for op in "vht":
        lst.extend([tf(t, op) for t in lst])
    with torch.no_grad():
        outs = [m.net_g(a) for a in lst]
    for i in range(8):
        if i > 3:
            outs[i] = tf(outs[i], "t")
        if i % 4 > 1:
            outs[i] = tf(outs[i], "h")
        if (i % 4) % 2 == 1:
            outs[i] = tf(outs[i], "v")
    want = torch.cat([o.unsqueeze(0) for o in outs], 0).mean(dim=0)
    assert torch.allclose(m.output, want, atol=1e-6)
    m.test()
    assert not torch.allclose(m.output, want, atol=1e-4)   # the conv is not flip-symmetric: the ensemble really differs

    class _Loader(list):
        dataset = type("D", (), {"opt": {"name": "v"}})()

    one = lq[:1]
    res_e = m.nondist_validation(_Loader([{"lq": one, "gt": (one * 0.9), "lq_path": ["a.png"]}]), 1, None, False)
    m2 = build_model(_opt(network_g=dict(type="_TestConvArch")))
    res_p = m2.nondist_validation(_Loader([{"lq": one, "gt": (one * 0.9), "lq_path": ["a.png"]}]), 1, None, False)
    m.feed_data({"lq": one})
    m.test_selfensemble()
    from basicsr.metrics import calculate_psnr
    direct = calculate_psnr(m.output.clamp(0, 1).numpy(), (one * 0.9).numpy(), crop_border=0, test_y_channel=False)
    assert abs(res_e["psnr"] - direct) < 1e-9 and abs(res_e["psnr"] - res_p["psnr"]) > 1e-6


# ------------------------------------------------------------------------------------------------
def _bucket_order_worker(rank, world, port, out):
    import torch.distributed as dist
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from torch.nn.parallel import DistributedDataParallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from basicsr.archs import build_network
    from oracle import nafnet_oracle as O

    torch.manual_seed(0)
    net = build_network(dict(type="NAFNetBaseline", img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1]))
    with torch.no_grad():
        for k, p in net.named_parameters():   # (beta / gamma = 0 would cut the graph behind every block)
            if k.endswith("beta") or k.endswith("gamma"):
                p.fill_(0.1)

    class OnCpu(torch.nn.Module):   # the product network's module tree and parameters, the oracle's arithmetic (the HIP kernels need a GPU)
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, x):
            return O.nafnet_forward(x, dict(self.net.named_parameters()))[0]

    names = {id(p): k for k, p in net.named_parameters()}
    model = DistributedDataParallel(OnCpu(net), bucket_cap_mb=64, gradient_as_bucket_view=True)
    buckets, ready = [], []

    def hook(state, bucket):
        buckets.append([names[id(p)] for p in bucket.parameters()])
        return default_hooks.allreduce_hook(state, bucket)

    model.register_comm_hook(None, hook)
    for p in net.parameters():
        p.register_post_accumulate_grad_hook(lambda q: ready.append(names[id(q)]))
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.rand(1, 3, 16, 16, generator=g)
    for it in range(3):
        buckets.clear()
        ready.clear()
        model.zero_grad(set_to_none=True)
        model(x).abs().mean().backward()
    torch.save({"buckets": list(buckets), "ready": list(ready)}, os.path.join(out, f"b{rank}.pt"))
    dist.destroy_process_group()


def test_ddp_bucket_order_is_backward_completion_order_gloo(tmp_path):
    """SURVEY 8e / reference base_model.py:108-115: NAFNet-64 under DDP with the 64 MB buckets of bench.py and BaseModel.model_to_device,
    two ranks on gloo.  After DDP's one re-bucketing the buckets hold the parameters in the order the backward pass finishes them and are
    all-reduced in that order -- ending first, then decoder3 ... decoder0, middle, the 28 blocks of encoders.3 from last to first, ...,
    intro last -- so every bucket's all-reduce has the rest of the backward pass to hide behind."""
    import torch.multiprocessing as mp

    port = 23500 + (os.getpid() % 2000)
    mp.spawn(_bucket_order_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"b{i}.pt")) for i in range(2)]
    assert r[0]["buckets"] == r[1]["buckets"]
    buckets, ready = r[0]["buckets"], r[0]["ready"]
    flat = [k for b in buckets for k in b]
    assert sorted(flat) == sorted(ready) and len(flat) == 664
    assert len(buckets) >= 4                      # 271.6 MB of gradients in 64 MB buckets
    # bucket k holds exactly the k-th run of the completion order
    pos = {k: i for i, k in enumerate(ready)}
    lo = 0
    for b in buckets:
        idx = sorted(pos[k] for k in b)
        assert idx == list(range(lo, lo + len(b))), (lo, idx[:4])
        lo += len(b)
    group = lambda k: k.split(".")[0] if not k.startswith("encoders") else ".".join(k.split(".")[:2])   # noqa: E731
    seen = []
    for k in ready:
        if not seen or seen[-1] != group(k):
            seen.append(group(k))
    blocks_only = [s for s in seen if not s.startswith(("ups", "downs"))]
    assert blocks_only == ["ending", "decoder3", "decoder2", "decoder1", "decoder0", "middle_blks", "encoders.3", "encoders.2", "encoders.1",
                           "encoders.0", "intro"], blocks_only
    enc3 = [int(k.split(".")[2]) for k in ready if k.startswith("encoders.3.") and k.endswith("conv1.weight")]
    assert enc3 == list(range(27, -1, -1))


# ------------------------------------------------------------------------------------------------
def _ddp_fallback_worker(rank, world, port, out):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from dcpt_amd import ddp as DD, functional as DF

    class _Emu(torch.autograd.Function):
        """stands in for a fused HIP block on the CPU: y = x w^T + b with its parameter gradients written into the buffers
        functional._grad_buffers hands out -- DDP's bucket views where they are known and unused, fresh tensors otherwise"""

        @staticmethod
        def forward(ctx, x, w, b):
            ctx.owners = (w, b)
            ctx.save_for_backward(x, w.detach(), b.detach())
            return x @ w.detach().t() + b.detach()

        @staticmethod
        def backward(ctx, dy):
            x, w, b = ctx.saved_tensors
            gw, gb = DF._grad_buffers((w, b), ctx.owners)
            gw.copy_(dy.t() @ x)
            gb.copy_(dy.sum(0))
            return dy @ w, gw, gb

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(3)
            self.w1 = torch.nn.Parameter(torch.randn(16, 8, generator=g))
            self.b1 = torch.nn.Parameter(torch.randn(16, generator=g))
            self.w2 = torch.nn.Parameter(torch.randn(4, 16, generator=g))
            self.b2 = torch.nn.Parameter(torch.randn(4, generator=g))

        def forward(self, x):
            return _Emu.apply(torch.tanh(_Emu.apply(x, self.w1, self.b1)), self.w2, self.b2)

    class NoBuiltin(DistributedDataParallel):   # a torch without the private hook registration dcpt_amd.ddp prefers
        def __getattribute__(self, name):
            if name == "_register_builtin_comm_hook":
                raise AttributeError(name)
            return super().__getattribute__(name)

    g = torch.Generator().manual_seed(100 + rank)
    xa, xb = torch.randn(6, 8, generator=g), torch.randn(6, 8, generator=g)
    res = {}
    for variant in ("builtin", "fallback"):
        net = Net()
        model = (DistributedDataParallel if variant == "builtin" else NoBuiltin)(net, gradient_as_bucket_view=True)
        assert hasattr(model, "_register_builtin_comm_hook") == (variant == "builtin")
        DD.prepare(model)
        opt = torch.optim.SGD(net.parameters(), lr=0.0)   # lr 0: the parameters stay put, the post-step hook still records the bucket views
        hits0 = DF._grad_buffers.hits
        grads = []
        for it in range(3):   # iteration 0: fresh tensors (no views known yet); from 1 on: the kernels' outputs ARE the bucket views
            opt.zero_grad(set_to_none=True)
            model(xa).square().mean().backward()
            grads.append([p.grad.clone() for p in net.parameters()])
            opt.step()
        res[variant] = dict(grads=grads, hits=DF._grad_buffers.hits - hits0)
        # gradient accumulation: micro-batch a under no_sync(), micro-batch b synchronised -> mean over ranks of the SUM of both
        opt.zero_grad(set_to_none=True)
        with model.no_sync():
            model(xa).square().mean().backward()
        model(xb).square().mean().backward()
        res[variant]["accum"] = [p.grad.clone() for p in net.parameters()]
    # what the ranks should have: plain autograd on an unwrapped copy
    ref = Net()
    own = {}
    for name, x in (("a", xa), ("b", xb)):
        ref.zero_grad(set_to_none=True)
        ref(x).square().mean().backward()
        own[name] = [p.grad.clone() for p in ref.parameters()]
    torch.save(dict(res=res, own=own), os.path.join(out, f"f{rank}.pt"))
    dist.destroy_process_group()


def test_ddp_zero_copy_gradients_fallback_hook_and_no_sync_gloo(tmp_path):
    """dcpt_amd/ddp.py on 2 gloo ranks (reference base_model.py:108-115 wraps the network in DDP): (1) with torch's private
    ``_register_builtin_comm_hook`` ABSENT ``prepare`` falls back to the public Python all-reduce hook and the averaged gradients are the
    same; (2) gradients written straight into DDP's bucket views (functional._grad_buffers, from the second iteration on) equal the mean
    of the ranks' own gradients; (3) accumulation over two micro-batches, the first under ``no_sync()``, equals the mean over ranks of the
    sum of both micro-batches' gradients -- a view is used at most once between two optimizer steps, the second backward falls back to
    a fresh tensor that autograd accumulates."""
    import torch.multiprocessing as mp

    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_ddp_fallback_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"f{i}.pt")) for i in range(2)]
    mean_a = [(x + y) / 2 for x, y in zip(r[0]["own"]["a"], r[1]["own"]["a"])]
    mean_ab = [(xa + xb + ya + yb) / 2 for xa, xb, ya, yb in zip(r[0]["own"]["a"], r[0]["own"]["b"], r[1]["own"]["a"], r[1]["own"]["b"])]
    for rank in range(2):
        for variant in ("builtin", "fallback"):
            v = r[rank]["res"][variant]
            for it in range(3):
                for g, want in zip(v["grads"][it], mean_a):
                    assert torch.allclose(g, want, rtol=1e-6, atol=1e-7), (rank, variant, it)
            assert v["hits"] >= 4, (variant, v["hits"])   # the zero-copy path was taken (4 parameters, iterations 1 and 2)
            for g, want in zip(v["accum"], mean_ab):
                assert torch.allclose(g, want, rtol=1e-6, atol=1e-7), (rank, variant, "no_sync accumulation")
        for a, b in zip(r[rank]["res"]["builtin"]["grads"][2], r[rank]["res"]["fallback"]["grads"][2]):
            assert torch.equal(a, b)


def test_pack_generation_follows_only_optimizers_that_own_packed_parameters():
    """dcpt_amd.functional's optimizer post-step hook: the weight-pack generation moves when an optimizer steps over a parameter some pack
    was made from (tagged by PackedWeightsBf16.tag) -- not for an unrelated optimizer (round-4 advisor: in the DCPT step optimizer_dc.step()
    made the frozen encoder repack every iteration)."""
    import torch
    from dcpt_amd import functional as DF

    names = DF._PACK_DEPS
    blockp = {k: torch.nn.Parameter(torch.zeros(2)) for k in names}
    other = torch.nn.Parameter(torch.zeros(2))
    DF.PackedWeightsBf16.tag(blockp)
    for p in list(blockp.values()) + [other]:
        p.grad = torch.ones_like(p)
    o_other, o_block = torch.optim.SGD([other], lr=0.1), torch.optim.SGD(list(blockp.values()), lr=0.1)
    g0 = DF._PACK_GENERATION
    o_other.step()
    assert DF._PACK_GENERATION == g0
    o_block.step()
    assert DF._PACK_GENERATION == g0 + 1
    assert DF.invalidate_packed_weights() == g0 + 2   # (EMA updates and checkpoint loads still invalidate explicitly)


def test_conv_pack_cache_host_logic():
    """dcpt_amd.functional.PackedConvBf16 (ABI 14, the bf16 head's cached operand images) without a GPU: the key follows the generation
    counter, the parameter's ``_version`` and its address; a copied / pickled module starts with an empty cache (no device event is
    copied); the head registers exactly its 34 convs; with the cache switched off nothing is packed and nothing is touched; a CPU weight
    is refused by the pack call (no CPU fallback)."""
    import copy
    import pickle

    import pytest
    import torch
    from basicsr.archs.degrad_classify_arch import PromptIR_NoImg_DC
    from dcpt_amd import _lib
    from dcpt_amd import functional as DF

    w = torch.nn.Parameter(torch.zeros(8, 8, 1, 1))
    k0 = DF.PackedConvBf16.key_of(w)
    with torch.no_grad():
        w.add_(1.0)
    k1 = DF.PackedConvBf16.key_of(w)
    assert k1 != k0 and k1[0] == k0[0]            # the version moved, the generation did not
    DF.invalidate_packed_weights()
    assert DF.PackedConvBf16.key_of(w)[0] == k0[0] + 1
    pk = DF.PackedConvBf16()
    pk.key, pk.buf = k1, torch.zeros(4)
    for c in (copy.deepcopy(pk), pickle.loads(pickle.dumps(pk))):
        assert c.key is None and c.buf is None and c.event is None
    head = PromptIR_NoImg_DC([64, 128, 256, 512], 2, 10, act_dtype="bf16")
    convs = head._packed_convs()
    assert len(convs) == 34 and len({id(c) for c, _ in convs}) == 34
    assert all(wt.dim() == 4 and wt.shape[2] in (1, 3) for _, wt in convs)
    ema = copy.deepcopy(head)
    assert all(a is not b for (a, _), (b, _) in zip(convs, ema._packed_convs()))
    lib = _lib.load()
    for _, wt in convs:   # the buffer holds both images, each rounded up to 256 bytes
        Co, Ci, ks = wt.shape[0], wt.shape[1], wt.shape[2]
        half = (Co * Ci * ks * ks * 2 + 255) // 256 * 256
        assert lib.dcpt_conv_wpack_bf16_bytes(Ci, Co, ks) == 2 * half
    old = DF.CONV_PACK_CACHE
    try:
        DF.CONV_PACK_CACHE = False
        assert DF.pack_convs_bf16(convs) == 0 and all(c.key is None for c, _ in convs)
        assert DF._pk(convs[0][0], convs[0][1]) == (None, 0)
        DF.CONV_PACK_CACHE = True
        with pytest.raises(_lib.DcptHipError):
            DF.pack_convs_bf16(convs[:2])          # CPU tensors: refused
    finally:
        DF.CONV_PACK_CACHE = old
