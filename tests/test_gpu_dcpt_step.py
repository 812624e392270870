"""GPU parity of the DCPT step (M1): basicsr.models DCPTModel / DCTModel / DCModel ``optimize_parameters`` on the HIP
path vs the golden re-enactment with the real reference archs and vs the oracle."""
import os

import numpy as np
import pytest
import torch

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict
from oracle import dc_oracle as D
from oracle import nafnet_oracle as O

pytestmark = pytest.mark.gpu
TINY = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])
DC_CFG = dict(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10)


def _opt(model_type):
    return dict(name="t", model_type=model_type, scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
                hook_names="decoder", network_g=dict(type="NAFNetBaseline", **TINY),
                network_dc=dict(type="PromptIR_NoImg_DC", **DC_CFG), path=dict(),
                train=dict(pixel_opt=dict(type="L1Loss", loss_weight=1.0, reduction="mean"),
                           classify_opt=dict(type="CrossEntropyLoss", loss_weight=1.0),
                           optim_g=dict(type="SGD", lr=0.0), optim_dc=dict(type="SGD", lr=0.0)))


def _build(model_type):
    from basicsr.models import build_model

    m = build_model(_opt(model_type))
    m.net_g.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
    m.net_dc.load_state_dict(keyed_state_dict(D.dc_param_shapes(**DC_CFG), seed=0), strict=True)
    return m


def _oracle_step(recon_on_lq, freeze):
    Pg = {k: v.clone().requires_grad_(not freeze) for k, v in keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0).items()}
    Pd = {k: v.clone().requires_grad_(True) for k, v in keyed_state_dict(D.dc_param_shapes(**DC_CFG), seed=0).items()}
    gt, lq = keyed_input("dcpt.gt", (2, 3, 32, 32)), keyed_input("dcpt.lq", (2, 3, 32, 32))
    total, l_pix = 0, None
    if not freeze:
        pix, _ = O.nafnet_forward(lq if recon_on_lq else gt, Pg)
        l_pix = O.l1_loss(pix, gt)
        total = total + l_pix
    _, taps = O.nafnet_forward(lq, Pg, hook=True)
    if freeze:
        taps = [t.detach() for t in taps]
    l_cls = torch.nn.functional.cross_entropy(D.dc_forward(taps[::-1], Pd), torch.tensor([3, 8]))
    (total + l_cls).backward()
    return l_pix, l_cls, Pg, Pd


def _feed(m):
    m.feed_data({"lq": keyed_input("dcpt.lq", (2, 3, 32, 32)), "gt": keyed_input("dcpt.gt", (2, 3, 32, 32)),
                 "dataset_idx": torch.tensor([3, 8])})


def test_dcpt_step_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "dcpt_step.npz"))
    m = _build("DCPTModel")
    assert len(m.hooks) == 4
    _feed(m)
    m.optimize_parameters(1)
    log = m.get_current_log()
    assert abs(log["l_pix"] - float(g["l_pix"])) < 1e-5 and abs(log["l_classify"] - float(g["l_classify"])) < 1e-4
    assert np.abs(m.cls_output.cpu().numpy() - g["logits"]).max() < 1e-3 * np.abs(g["logits"]).max()
    for tag, net in (("g", m.net_g), ("dc", m.net_dc)):
        params = dict(net.named_parameters())
        for n, l2 in zip([str(s) for s in g[f"{tag}_names"]], g[f"{tag}_l2"]):
            mine = float(params[n].grad.double().pow(2).sum().sqrt())
            assert abs(mine - l2) <= 2e-3 * max(1e-7, l2), (tag, n, mine, l2)
    for k in ("g.intro.weight", "g.ending.weight", "g.decoder3.0.conv5.weight", "dc.fc.weight", "dc.mixing_weights"):
        tag, name = k.split(".", 1)
        mine = dict((m.net_g if tag == "g" else m.net_dc).named_parameters())[name].grad.cpu().numpy()
        assert np.abs(mine - g[k]).max() <= 1e-3 * np.abs(g[k]).max(), k
    assert m.hook_outputs == []


@pytest.mark.parametrize("model_type,recon_on_lq,freeze", [("DCTModel", True, False), ("DCModel", False, True)])
def test_variants_vs_oracle(model_type, recon_on_lq, freeze):
    l_pix, l_cls, Pg, Pd = _oracle_step(recon_on_lq, freeze)
    m = _build(model_type)
    _feed(m)
    m.optimize_parameters(1)
    log = m.get_current_log()
    assert abs(log["l_classify"] - float(l_cls)) < 1e-4
    if not freeze:
        assert abs(log["l_pix"] - float(l_pix)) < 1e-5
    for name, p in m.net_dc.named_parameters():
        ref = Pd[name].grad
        assert float((p.grad.cpu() - ref).abs().max()) <= 2e-3 * max(1e-7, float(ref.abs().max())), name
    for name, p in m.net_g.named_parameters():
        if freeze:
            assert p.grad is None
        else:
            ref = Pg[name].grad
            assert float((p.grad.cpu() - ref).abs().max()) <= 2e-3 * max(1e-7, float(ref.abs().max())), name
    m.test()
    assert m.cls_output.shape == (2, 10)


@pytest.mark.parametrize("model_type", ["DCPTModel", "DCTModel"])
def test_batched_encoder_pass_equals_two_passes(model_type):
    """train.batched_encoder_passes (default on): one encoder pass over [reconstruction input; lq] instead of the reference's two
    passes -- same losses, logits and gradients up to fp32 summation order (the golden / oracle tests above run the batched form
    against the real reference; this one pins the two forms of THIS repo to each other)."""
    res = {}
    for batched in (True, False):
        from basicsr.models import build_model

        opt = _opt(model_type)
        opt["train"]["batched_encoder_passes"] = batched
        m = build_model(opt)
        m.net_g.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
        m.net_dc.load_state_dict(keyed_state_dict(D.dc_param_shapes(**DC_CFG), seed=0), strict=True)
        assert m.batched_encoder_passes is batched
        _feed(m)
        m.optimize_parameters(1)
        log = m.get_current_log()
        res[batched] = (log, m.cls_output.cpu(), {"g." + k: p.grad.cpu() for k, p in m.net_g.named_parameters()} |
                        {"dc." + k: p.grad.cpu() for k, p in m.net_dc.named_parameters()})
        assert m.hook_outputs == []
    (la, ca, ga), (lb, cb, gb) = res[True], res[False]
    assert abs(la["l_pix"] - lb["l_pix"]) < 1e-6 and abs(la["l_classify"] - lb["l_classify"]) < 1e-5
    assert float((ca - cb).abs().max()) <= 1e-5 * float(cb.abs().max())
    for k in gb:
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-5 * max(1e-7, float(gb[k].abs().max())), k


# ------------------------------------------------------------------------------------------------
DIST_G = dict(dim=16, num_blocks=[4, 6, 6, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8])
DIST_DC = dict(feature_dims=[32, 32, 64], num_res_blocks=1, num_classes=5)


def _dist_model(**over):
    from basicsr.archs import build_network
    from basicsr.models import build_model

    opt = dict(name="t", model_type="DCDistModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
               hook_names="decoder_level", network_g=dict(type="Restormer_origin", **DIST_G),
               network_dc=dict(type="PromptIR_NoImg_DC", **DIST_DC), path=dict(),
               train=dict(pixel_opt=dict(type="L1Loss", loss_weight=1.0, reduction="mean"),
                          classify_opt=dict(type="CrossEntropyLoss", loss_weight=1.0), optim_g=dict(type="SGD", lr=0.0)))
    opt.update(over)
    m = build_model(opt)
    shapes = {k: tuple(v.shape) for k, v in build_network(dict(type="Restormer_origin", **DIST_G)).state_dict().items()}
    m.net_g.load_state_dict(keyed_state_dict(shapes, seed=0), strict=True)
    m.net_dc.load_state_dict(keyed_state_dict(D.dc_param_shapes(**DIST_DC), seed=0), strict=True)
    if hasattr(m, "net_g_ema"):
        m.model_ema(0)   # re-copy: the EMA net was initialised from the constructor's weights
    return m


def test_dcdist_step_golden(golden_dir):
    """DCDistModel.optimize_parameters (reference ..._distillation_model.py:152-185) on the HIP path vs the re-enactment with
    the real reference archs: one net_g forward, taps = last block of each decoder level, frozen head, one backward"""
    from basicsr.models.degradation_classification_distillation_model import tap_modules

    g = np.load(os.path.join(golden_dir, "dcdist_step.npz"))
    m = _dist_model()
    assert [n for n, _ in tap_modules(m.net_g, "decoder_level")] == [str(s) for s in g["hooked"]]
    assert len(m.hooks) == 3 and not m.net_dc.training and all(not p.requires_grad for p in m.net_dc.parameters())
    m.feed_data({"lq": keyed_input("dist.lq", (2, 3, 32, 32)), "gt": keyed_input("dist.gt", (2, 3, 32, 32)),
                 "dataset_idx": torch.tensor([4, 1])})
    m.optimize_parameters(1)
    log = m.get_current_log()
    assert abs(log["l_pixel"] - float(g["l_pixel"])) < 1e-5 and abs(log["l_classify"] - float(g["l_classify"])) < 1e-4
    assert np.abs(m.cls_output.detach().cpu().numpy() - g["logits"]).max() < 1e-3 * np.abs(g["logits"]).max()
    assert np.abs(m.pix_output.detach().cpu().numpy() - g["pix"]).max() < 1e-4 * np.abs(g["pix"]).max()
    params = dict(m.net_g.named_parameters())
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(params[n].grad.double().pow(2).sum().sqrt())
        # a temperature gradient is ONE scalar summed over ReLU-masked products with heavy cancellation: in this network the
        # reference's own fp32 value differs from an fp64 evaluation by up to 1e-2 relative (4e-4 absolute), so it gets an
        # absolute allowance on top of the relative one
        slack = 1e-3 if n.endswith(".temperature") else 0.0
        assert abs(mine - l2) <= 2e-3 * max(1e-7, l2) + slack, (n, mine, l2)
    # full gradients: this 34-block network amplifies fp32 rounding (ReLU attention masks flip), the reference's own fp32
    # gradients are up to 2e-3 (scale-relative) away from an fp64 evaluation -- so measure both against the fp64 oracle and
    # require the HIP path to be as accurate as the reference (2x + 1e-4)
    from oracle import restormer_oracle as R

    P64 = {k: v.double().requires_grad_(True) for k, v in m.net_g.state_dict().items()}
    Pd64 = {k: v.double() for k, v in m.net_dc.state_dict().items()}
    P64 = {k: v.detach().cpu().requires_grad_(True) for k, v in P64.items()}
    Pd64 = {k: v.cpu() for k, v in Pd64.items()}
    lq, gt = keyed_input("dist.lq", (2, 3, 32, 32)).double(), keyed_input("dist.gt", (2, 3, 32, 32)).double()
    pix, taps = R.restormer_forward(lq, P64, origin=True)
    ((pix - gt).abs().mean() + torch.nn.functional.cross_entropy(D.dc_forward(taps[::-1], Pd64), torch.tensor([4, 1]))).backward()
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            truth = P64[k[2:]].grad.numpy()
            scale = np.abs(truth).max()
            err_ref = np.abs(g[k] - truth).max() / scale
            err_hip = np.abs(params[k[2:]].grad.cpu().numpy() - truth).max() / scale
            assert err_hip <= 2 * err_ref + 1e-4, (k, err_hip, err_ref)
    assert all(p.grad is None for p in m.net_dc.parameters()) and m.hook_outputs == []


def test_dcdist_training_options():
    """constant ``dataset_idx`` from the options, gradient clipping, EMA, and the SRModel-style test path"""
    m = _dist_model(dataset_idx=3, grad_clip=0.01,
                    train=dict(pixel_opt=dict(type="L1Loss", loss_weight=1.0, reduction="mean"),
                               classify_opt=dict(type="CrossEntropyLoss", loss_weight=1.0), optim_g=dict(type="SGD", lr=0.1),
                               ema_decay=0.9))
    before = {k: v.detach().clone() for k, v in m.net_g.named_parameters()}
    m.feed_data({"lq": keyed_input("dist.lq", (2, 3, 32, 32)), "gt": keyed_input("dist.gt", (2, 3, 32, 32))})
    assert m.dataset_idx.tolist() == [3, 3]
    m.optimize_parameters(1)
    sq = sum(float((p.detach() - before[k]).double().pow(2).sum()) for k, p in m.net_g.named_parameters())
    assert 0 < sq ** 0.5 <= 0.1 * 0.01 * 1.0001   # |step| = lr * clipped gradient norm
    ema = dict(m.net_g_ema.named_parameters())
    for k, p in m.net_g.named_parameters():   # ema = 0.9 * old + 0.1 * new (model_ema(0) copied the initial weights)
        want = 0.9 * before[k] + 0.1 * p.detach()
        assert float((ema[k].detach() - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), k
    m.feed_data({"lq": keyed_input("dist.lq", (1, 3, 30, 27)), "gt": keyed_input("dist.gt", (1, 3, 30, 27))})
    m.opt["network_g"]["window_size"] = 8
    m.pre_test(); m.test(); m.post_test()
    assert tuple(m.output.shape) == (1, 3, 30, 27) and m.hook_outputs == []
    with pytest.raises(ValueError):
        _dist_model(train=dict(optim_g=dict(type="SGD", lr=0.0)))


# ------------------------------------------------------------------------------------------------------------------------
def _dcpt_two_rank_worker(rank, world, port, out, batched, act, gemm_mode):
    """one data-parallel rank of the DCPT step with the HIP networks (both wrapped in DDP by BaseModel.model_to_device): the two ranks share
    cuda:0 (there is one GPU), the collectives go over gloo"""
    import sys

    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from torch.nn.parallel import DistributedDataParallel

    if gemm_mode != "fp32":   # (the suite's second pass: the wide GEMMs as split-operand bf16x3 products, as the parent process runs them)
        from dcpt_amd import functional as DF

        DF.set_gemm_precision(gemm_mode, min_tiles=1, scratch_mb=768)
    opt = _opt("DCPTModel")
    opt.update(dist=True, rank=rank, world_size=world, ddp_bucket_cap_mb=0.05)
    opt["train"]["batched_encoder_passes"] = batched
    opt["network_g"]["act_dtype"] = act
    opt["network_dc"]["act_dtype"] = "bf16" if act != "fp32" else "fp32"
    from basicsr.models import build_model

    m = build_model(opt)
    assert isinstance(m.net_g, DistributedDataParallel) and isinstance(m.net_dc, DistributedDataParallel)
    m.get_bare_model(m.net_g).load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
    m.get_bare_model(m.net_dc).load_state_dict(keyed_state_dict(D.dc_param_shapes(**DC_CFG), seed=0), strict=True)
    grads = None
    for it in range(3):   # (DDP re-buckets after its first iteration; the third one runs on the final buckets with the views remembered)
        m.feed_data({"lq": keyed_input(f"dcpt2r.lq{rank}", (2, 3, 32, 32)), "gt": keyed_input(f"dcpt2r.gt{rank}", (2, 3, 32, 32)),
                     "dataset_idx": torch.tensor([3 + rank, 8 - rank])})
        m.optimize_parameters(it + 1)
        torch.cuda.synchronize()
        grads = {tag: {k: p.grad.detach().float().cpu().clone() for k, p in m.get_bare_model(net).named_parameters()}
                 for tag, net in (("g", m.net_g), ("dc", m.net_dc))}
    torch.save({"grads": grads, "log": m.get_current_log()}, os.path.join(out, f"dcpt{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("batched,act", [(True, "fp32"), (False, "fp32"), (True, "bf16")])
def test_dcpt_step_two_ranks_one_gpu_gloo(tmp_path, batched, act, gemm_mode):
    """SURVEY 8e's "DCPT quirk" with the HIP networks: TWO data-parallel ranks of the DCPT step (reference ...pretrain_model.py:133-169 under
    base_model.py:108-115: encoder AND head each wrapped in DDP), both on cuda:0, the all-reduce over gloo.  batched = the stacked encoder pass
    (one forward of net_g over 2B samples with the head's taps split off inside the graph); not batched = the reference's two forwards of
    net_g before ONE backward, where a gradient bucket is ready only after both uses of every weight have back-propagated.  Every rank must
    end with the MEAN of the two ranks' single-process gradients for both networks, and the two ranks must agree bit for bit."""
    import torch.multiprocessing as mp

    port = 26500 + (os.getpid() % 2000)
    mp.spawn(_dcpt_two_rank_worker, args=(2, port, str(tmp_path), batched, act, gemm_mode), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"dcpt{i}.pt")) for i in range(2)]
    local = []
    for rank in range(2):
        from basicsr.models import build_model

        opt = _opt("DCPTModel")
        opt["train"]["batched_encoder_passes"] = batched
        opt["network_g"]["act_dtype"] = act
        opt["network_dc"]["act_dtype"] = "bf16" if act != "fp32" else "fp32"
        m = build_model(opt)
        m.net_g.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
        m.net_dc.load_state_dict(keyed_state_dict(D.dc_param_shapes(**DC_CFG), seed=0), strict=True)
        m.feed_data({"lq": keyed_input(f"dcpt2r.lq{rank}", (2, 3, 32, 32)), "gt": keyed_input(f"dcpt2r.gt{rank}", (2, 3, 32, 32)),
                     "dataset_idx": torch.tensor([3 + rank, 8 - rank])})
        m.optimize_parameters(1)
        local.append({tag: {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()} for tag, net in (("g", m.net_g), ("dc", m.net_dc))})
    tol = 1e-6 if act == "fp32" else 1e-5   # (the mean of two fp32 tensors, computed by gloo on the host vs here)
    for tag in ("g", "dc"):
        assert list(r[0]["grads"][tag]) == list(local[0][tag])
        for k in local[0][tag]:
            want = (local[0][tag][k] + local[1][tag][k]) / 2
            assert torch.equal(r[0]["grads"][tag][k], r[1]["grads"][tag][k]), (tag, k)
            assert torch.allclose(r[0]["grads"][tag][k], want, rtol=tol, atol=tol * float(want.abs().max()) + 1e-12), (tag, k)
