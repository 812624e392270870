"""ORACLE tooling: generate tests/golden/*.npz from the REAL reference (build container only).

    python oracle/make_golden.py

Imports the reference archs from /root/reference (oracle/ref_import.py), loads the keyed
deterministic weights (dcpt_amd/keyed_init.py) into them with ``load_state_dict(strict=True)``
and records outputs + gradients.  The fixtures are data only (inputs are regenerated from
keys; outputs/gradients are stored); no reference source is stored.  The reference publishes
no golden vectors of its own, so these are the pins.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dcpt_amd.keyed_init import fill_module_, keyed_input, keyed_tensor  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def _grad_summary(module):
    names, l2, s, a = [], [], [], []
    for k, p in module.named_parameters():
        g = p.grad.detach().double()
        names.append(k)
        l2.append(float(g.pow(2).sum().sqrt()))
        s.append(float(g.sum()))
        a.append(float(g.abs().sum()))
    return np.array(names), np.array(l2), np.array(s), np.array(a)


def gen_ln(ref):
    out = {}
    for tag, shape in [("a", (2, 64, 8, 8)), ("b", (1, 512, 4, 4)), ("c", (3, 8, 5, 7))]:
        C = shape[1]
        x = keyed_input(f"ln.{tag}.x", shape, lo=-2.0, hi=3.0).requires_grad_(True)
        w = keyed_tensor(f"ln.{tag}.norm.weight", (C,)).requires_grad_(True)
        b = keyed_tensor(f"ln.{tag}.norm.bias", (C,)).requires_grad_(True)
        go = keyed_input(f"ln.{tag}.go", shape, lo=-1.0, hi=1.0)
        y = ref.nafnet.LayerNormFunction.apply(x, w, b, 1e-6)
        y.backward(go)
        out[f"{tag}.y"] = _np(y)
        out[f"{tag}.dx"] = _np(x.grad)
        out[f"{tag}.dw"] = _np(w.grad)
        out[f"{tag}.db"] = _np(b.grad)
    np.savez_compressed(os.path.join(OUT, "ln2d.npz"), **out)


def gen_nafblock(ref):
    for c in (16, 64):
        blk = ref.nafnet.NAFBlock(c)
        # keys are prefixed so that different fixtures get different weights
        sd = {k: keyed_tensor(f"blk{c}." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
        blk.load_state_dict(sd, strict=True)
        x = keyed_input(f"blk{c}.x", (2, c, 16, 16), lo=-1.0, hi=1.0).requires_grad_(True)
        go = keyed_input(f"blk{c}.go", (2, c, 16, 16), lo=-1.0, hi=1.0)
        y = blk(x)
        y.backward(go)
        out = {"y": _np(y), "dx": _np(x.grad)}
        for k, p in blk.named_parameters():
            out["g." + k] = _np(p.grad)
        np.savez_compressed(os.path.join(OUT, f"nafblock_c{c}.npz"), **out)


TINY = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])
FULL = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])


def gen_nafnet_tiny(ref):
    net = ref.nafnet.NAFNetBaseline(**TINY)
    fill_module_(net, seed=0)
    x = keyed_input("tiny.x", (2, 3, 32, 32)).requires_grad_(True)
    gw = keyed_input("tiny.gw", (2, 3, 32, 32), lo=-1.0, hi=1.0)
    taps = []
    hooks = [getattr(net, f"decoder{i}").register_forward_hook(lambda m, i, o: taps.append(o)) for i in range(4)]
    y = net(x)
    (y * gw).sum().backward()
    out = {"y": _np(y), "dx": _np(x.grad)}
    for i, t in enumerate(taps):
        out[f"tap{i}"] = _np(t)
    names, l2, s, a = _grad_summary(net)
    out["g_names"], out["g_l2"], out["g_sum"], out["g_abs"] = names, l2, s, a
    for k in ("intro.weight", "intro.bias", "ending.weight", "ending.bias", "downs.0.weight", "downs.3.bias",
              "ups.0.0.weight", "ups.3.0.weight", "encoders.0.0.conv1.weight", "encoders.0.0.conv2.weight",
              "encoders.0.0.beta", "encoders.3.1.gamma", "middle_blks.0.sca.1.weight", "middle_blks.0.sca.1.bias",
              "decoder3.0.norm1.weight", "decoder3.0.norm2.bias", "decoder0.0.conv5.weight", "decoder2.0.conv3.bias"):
        out["g." + k] = _np(dict(net.named_parameters())[k].grad)
    # hook=True path returns None (nafnet_arch.py:269-274)
    for h in hooks:
        h.remove()
    assert net(x.detach(), hook=True) is None
    np.savez_compressed(os.path.join(OUT, "nafnet_tiny.npz"), **out)


def gen_nafnet_full(ref):
    """Full-size NAFNet-64 [1,1,1,28] (options/all_in_one/test/test_NAFNet_5d.yml:50-56) on one
    256x256 image: sub-sampled output + grad summaries (weights are regenerable from keys)."""
    torch.manual_seed(0)
    net = ref.nafnet.NAFNetBaseline(**FULL)
    fill_module_(net, seed=0)
    x = keyed_input("full.x", (1, 3, 256, 256)).requires_grad_(True)
    gt = keyed_input("full.gt", (1, 3, 256, 256))
    y = net(x)
    loss = (y - gt).abs().mean()
    loss.backward()
    names, l2, s, a = _grad_summary(net)
    out = {
        "y_sub": _np(y[..., ::16, ::16]),
        "y_mean": np.float64(y.double().mean().item()),
        "y_absmean": np.float64(y.double().abs().mean().item()),
        "loss": np.float64(loss.item()),
        "dx_sub": _np(x.grad[..., ::16, ::16]),
        "dx_l2": np.float64(x.grad.double().pow(2).sum().sqrt().item()),
        "g_names": names, "g_l2": l2, "g_sum": s, "g_abs": a,
        "n_params": np.int64(sum(p.numel() for p in net.parameters())),
        "n_keys": np.int64(len(net.state_dict())),
    }
    np.savez_compressed(os.path.join(OUT, "nafnet_full.npz"), **out)


def gen_tlsc(ref):
    pool = ref.arch_util.AvgPool2d(base_size=(24, 20), train_size=(1, 3, 32, 32), fast_imp=False)
    x = keyed_input("tlsc.x", (2, 8, 48, 40), lo=-1.0, hi=1.0)
    y = pool(x)
    np.savez_compressed(os.path.join(OUT, "tlsc.npz"), y=_np(y), kernel=np.array(pool.kernel_size))
    # NAFNet (Local_Base) end-to-end on a tiny config
    net = ref.nafnet.NAFNet(train_size=(1, 3, 16, 16), **TINY)
    fill_module_(net, seed=0)
    xi = keyed_input("tlsc.img", (1, 3, 48, 32))
    with torch.no_grad():
        yo = net(xi)
    np.savez_compressed(os.path.join(OUT, "nafnet_local_tiny.npz"), y=_np(yo))


DC_CFG = dict(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10)


def gen_dc_head(ref):
    """PromptIR_NoImg_DC (degrad_classify_arch.py:558-641) on four feature maps, CE loss, all gradients."""
    net = ref.dc.PromptIR_NoImg_DC(**DC_CFG)
    fill_module_(net, seed=0)
    feats = [keyed_input(f"dc.f{i}", (3, c, 32 >> i, 32 >> i), lo=-1.0, hi=1.0).requires_grad_(True)
             for i, c in enumerate(DC_CFG["feature_dims"])]
    labels = torch.tensor([1, 7, 4])
    logits = net(None, list(feats))
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out = {"logits": _np(logits), "loss": np.float64(loss.item())}
    for i, f in enumerate(feats):
        out[f"df{i}"] = _np(f.grad)
    names, l2, sm, ab = _grad_summary(net)
    out["g_names"], out["g_l2"], out["g_sum"], out["g_abs"] = names, l2, sm, ab
    for k, p in net.named_parameters():  # full gradients for the small tensors and the first / last blocks
        if p.numel() <= 4096 or k.startswith("bottleneck_layers.0.0.") or k.startswith("last_stage.1.conv1"):
            out["g." + k] = _np(p.grad)
    out["keys"] = np.array(list(net.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "dc_head.npz"), **out)


def gen_dc_img_head(ref):
    """PromptIR_DC (degrad_classify_arch.py:480-555): 7x7 stride-2 image embedding + LayerNorm, then the same stages; the
    feature maps live at H/2, H/4, ... of the image.  CE loss, gradients of the image, the features and the parameters.
    Also records that the reference rejects feature maps at the image resolution (NAFNet's taps)."""
    net = ref.dc.PromptIR_DC(**DC_CFG)
    fill_module_(net, seed=0)
    lq = keyed_input("dci.lq", (3, 3, 36, 28)).requires_grad_(True)
    # (36 x 28 embeds to 18 x 14, which MaxPool2d floors to 9 x 7 -> 4 x 3: that image only exercises the embedding;
    # the four-stage path uses a power-of-two image)
    lq2 = keyed_input("dci.lq2", (2, 3, 64, 32)).requires_grad_(True)
    feats2 = [keyed_input(f"dci.g{i}", (2, c, 32 >> i, 16 >> i), lo=-1.0, hi=1.0).requires_grad_(True)
              for i, c in enumerate(DC_CFG["feature_dims"])]
    labels = torch.tensor([5, 2])
    logits = net(lq2, list(feats2))
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out = {"logits": _np(logits), "loss": np.float64(loss.item()), "dlq": _np(lq2.grad)}
    for i, f in enumerate(feats2):
        out[f"df{i}"] = _np(f.grad)
    names, l2, sm, ab = _grad_summary(net)
    out["g_names"], out["g_l2"], out["g_sum"], out["g_abs"] = names, l2, sm, ab
    for k, p in net.named_parameters():
        if p.numel() <= 4096 or k.startswith("conv_embed."):
            out["g." + k] = _np(p.grad)
    out["keys"] = np.array(list(net.state_dict().keys()))
    # the embedding alone on an odd-sized image (36 x 28 -> 18 x 14), with its input gradient
    e = net.conv_embed(lq)
    ge = keyed_input("dci.ge", tuple(e.shape), lo=-1.0, hi=1.0)
    (dlq,) = torch.autograd.grad((e * ge).sum(), lq)
    out["embed"], out["embed_dlq"] = _np(e), _np(dlq)
    lq3 = keyed_input("dci.lq3", (1, 3, 37, 29))
    out["embed_odd"] = _np(net.conv_embed(lq3))
    try:  # features at the image resolution (what net_g's decoder taps have): shape error in the reference
        net(lq2.detach(), [keyed_input("dci.bad", (2, 8, 64, 32))] + [f.detach() for f in feats2[1:]])
        out["full_res_features_fail"] = np.array(False)
    except RuntimeError:
        out["full_res_features_fail"] = np.array(True)
    np.savez_compressed(os.path.join(OUT, "dc_img_head.npz"), **out)


def gen_dcpt_step(ref):
    """Re-enact DCPTModel.optimize_parameters (degradation_classification_pretrain_model.py:133-169) with the
    reference archs: net_g(gt) -> L1; net_g(lq, hook=True) with hooks on decoder{i}.0; net_dc(taps[::-1]) -> CE;
    one backward.  Stores both losses and every parameter-gradient norm of both nets."""
    net_g = ref.nafnet.NAFNetBaseline(**TINY)
    net_dc = ref.dc.PromptIR_NoImg_DC(**DC_CFG)
    fill_module_(net_g, seed=0)
    fill_module_(net_dc, seed=0)
    gt = keyed_input("dcpt.gt", (2, 3, 32, 32))
    lq = keyed_input("dcpt.lq", (2, 3, 32, 32))
    labels = torch.tensor([3, 8])
    taps = []
    for name, module in net_g.named_modules():
        if "decoder" in name and name.count(".") == 1:
            module.register_forward_hook(lambda m, i, o: taps.append(o))
    pix = net_g(gt, hook=False)
    taps.clear()
    l_pix = (pix - gt).abs().mean()
    assert net_g(lq, hook=True) is None
    assert len(taps) == 4
    cls = net_dc(lq, taps[::-1])
    l_cls = torch.nn.functional.cross_entropy(cls, labels)
    (l_pix + l_cls).backward()
    out = {"l_pix": np.float64(l_pix.item()), "l_classify": np.float64(l_cls.item()), "logits": _np(cls)}
    for tag, net in (("g", net_g), ("dc", net_dc)):
        names, l2, sm, ab = _grad_summary(net)
        out[f"{tag}_names"], out[f"{tag}_l2"], out[f"{tag}_sum"] = names, l2, sm
    out["g.intro.weight"] = _np(net_g.intro.weight.grad)
    out["g.ending.weight"] = _np(net_g.ending.weight.grad)
    out["g.decoder3.0.conv5.weight"] = _np(dict(net_g.named_parameters())["decoder3.0.conv5.weight"].grad)
    out["dc.fc.weight"] = _np(net_dc.fc.weight.grad)
    out["dc.mixing_weights"] = _np(net_dc.mixing_weights.grad)
    np.savez_compressed(os.path.join(OUT, "dcpt_step.npz"), **out)


DIST_G = dict(dim=16, num_blocks=[4, 6, 6, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8])
DIST_DC = dict(feature_dims=[32, 32, 64], num_res_blocks=1, num_classes=5)


def gen_dcdist_step(ref):
    """Re-enact DCDistModel.optimize_parameters (degradation_classification_distillation_model.py:152-185) with the reference
    archs: hooks on block 5 of decoder_level3/2 and block 3 of decoder_level1 of Restormer_origin (:81-88), ONE forward of
    net_g(lq), frozen eval net_dc on the reversed taps, L1 + CE, one backward.  Stores the losses, the logits, the restored
    image and every parameter-gradient norm of net_g."""
    net_g = ref.restormer.Restormer_origin(**DIST_G)
    net_dc = ref.dc.PromptIR_NoImg_DC(**DIST_DC)
    fill_module_(net_g, seed=0)
    fill_module_(net_dc, seed=0)
    net_dc.eval()
    for p in net_dc.parameters():
        p.requires_grad = False
    taps, hooked = [], []
    for name, module in net_g.named_modules():
        if "decoder_level" in name and name.count(".") == 1:
            pre, idx = name.split(".")[0], int(name.split(".")[-1])
            if (idx == 5 and int(pre[-1]) in [2, 3]) or (idx == 3 and int(pre[-1]) == 1):
                module.register_forward_hook(lambda m, i, o: taps.append(o))
                hooked.append(name)
    lq = keyed_input("dist.lq", (2, 3, 32, 32))
    gt = keyed_input("dist.gt", (2, 3, 32, 32))
    labels = torch.tensor([4, 1])
    pix = net_g(lq)
    assert len(taps) == 3
    cls = net_dc(lq, taps[::-1])
    l_pix = (pix - gt).abs().mean()
    l_cls = torch.nn.functional.cross_entropy(cls, labels)
    (l_pix + l_cls).backward()
    assert all(p.grad is None for p in net_dc.parameters())
    out = {"l_pixel": np.float64(l_pix.item()), "l_classify": np.float64(l_cls.item()), "logits": _np(cls), "pix": _np(pix),
           "hooked": np.array(hooked), "tap_shapes": np.array([list(t.shape) for t in taps])}
    names, l2, sm, ab = _grad_summary(net_g)
    out["g_names"], out["g_l2"], out["g_sum"] = names, l2, sm
    for k in ("patch_embed.proj.weight", "output.weight", "decoder_level1.3.ffn.project_out.weight", "decoder_level3.5.attn.temperature",
              "refinement.0.norm1.body.weight"):
        out["g." + k] = _np(dict(net_g.named_parameters())[k].grad)
    np.savez_compressed(os.path.join(OUT, "dcdist_step.npz"), **out)


R_CFG = dict(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8])


def gen_restormer(ref):
    """tiny Restormer (BiasFree LN, hook support) and Restormer_origin (WithBias LN): output, taps, dL/dinput, grad norms;
    plus a single TransformerBlock (dim 48, 1 head, hidden 127 = int(48*2.66)) with every gradient."""
    for tag, cls in (("restormer", ref.restormer.Restormer), ("restormer_origin", ref.restormer.Restormer_origin)):
        net = cls(**R_CFG)
        fill_module_(net, seed=0)
        x = keyed_input(f"{tag}.x", (2, 3, 32, 32)).requires_grad_(True)
        gw = keyed_input(f"{tag}.gw", (2, 3, 32, 32), lo=-1.0, hi=1.0)
        y = net(x)
        (y * gw).sum().backward()
        out = {"y": _np(y), "dx": _np(x.grad), "keys": np.array(list(net.state_dict().keys()))}
        names, l2, sm, ab = _grad_summary(net)
        out["g_names"], out["g_l2"], out["g_sum"] = names, l2, sm
        for k, p in net.named_parameters():
            if p.numel() <= 2048 and ("encoder_level1" in k or "latent" in k or "refinement" in k or k.startswith("output")):
                out["g." + k] = _np(p.grad)
        if tag == "restormer":
            assert net(x.detach(), hook=True) is None
        np.savez_compressed(os.path.join(OUT, f"{tag}_tiny.npz"), **out)
    for lnt in ("BiasFree", "WithBias"):
        blk = ref.restormer.TransformerBlock(dim=48, num_heads=1, ffn_expansion_factor=2.66, bias=False, LayerNorm_type=lnt)
        sd = {k: keyed_tensor(f"tb{lnt}." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
        blk.load_state_dict(sd, strict=True)
        x = keyed_input(f"tb{lnt}.x", (2, 48, 12, 10), lo=-1.0, hi=1.0).requires_grad_(True)
        go = keyed_input(f"tb{lnt}.go", (2, 48, 12, 10), lo=-1.0, hi=1.0)
        y = blk(x)
        y.backward(go)
        out = {"y": _np(y), "dx": _np(x.grad)}
        for k, p in blk.named_parameters():
            out["g." + k] = _np(p.grad)
        np.savez_compressed(os.path.join(OUT, f"restormer_block_{lnt}.npz"), **out)


P_CFG = dict(num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)   # dim 48 is hard-wired into the prompt sizes (:288-296)


def gen_promptir(ref):
    """PromptIR (basicsr/archs/promptir_arch.py:266-518), one block per level: output, dL/dinput, every parameter-gradient
    norm, full gradients of the prompt blocks / one noise block; a 40 x 24 image (prompts resized 16->5x3, 32->10x6,
    64->20x12: the downsampling side of the bilinear resize) forward only; hook=True returns None."""
    net = ref.promptir.PromptIR(**P_CFG)
    fill_module_(net, seed=0)
    x = keyed_input("pir.x", (2, 3, 64, 64)).requires_grad_(True)
    gw = keyed_input("pir.gw", (2, 3, 64, 64), lo=-1.0, hi=1.0)
    y = net(x)
    (y * gw).sum().backward()
    out = {"y": _np(y), "dx": _np(x.grad), "keys": np.array(list(net.state_dict().keys()))}
    names, l2, sm, ab = _grad_summary(net)
    out["g_names"], out["g_l2"], out["g_sum"] = names, l2, sm
    for k, p in net.named_parameters():
        if (k.startswith("prompt") and p.numel() <= 65536) or k.startswith("noise_level1.norm1") or \
                k in ("noise_level1.attn.temperature", "noise_level1.attn.qkv_dwconv.weight", "noise_level3.attn.temperature"):
            out["g." + k] = _np(p.grad)
    out["g.prompt1.prompt_param.sub"] = _np(net.prompt1.prompt_param.grad[0, :, ::8, ::4, ::4])
    assert net(x.detach(), hook=True) is None
    with torch.no_grad():
        out["y_small"] = _np(net(keyed_input("pir.xs", (1, 3, 40, 24))))
        out["y_large"] = _np(net(keyed_input("pir.xl", (1, 3, 160, 136)))[..., ::4, ::4])
    np.savez_compressed(os.path.join(OUT, "promptir_tiny.npz"), **out)
    # one PromptIR transformer block (softmax attention, eps 1e-5) with every gradient, both LayerNorm types
    for lnt in ("BiasFree", "WithBias"):
        blk = ref.promptir.TransformerBlock(dim=48, num_heads=2, ffn_expansion_factor=2.66, bias=False, LayerNorm_type=lnt)
        sd = {k: keyed_tensor(f"ptb{lnt}." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
        blk.load_state_dict(sd, strict=True)
        xb = keyed_input(f"ptb{lnt}.x", (2, 48, 12, 10), lo=-1.0, hi=1.0).requires_grad_(True)
        go = keyed_input(f"ptb{lnt}.go", (2, 48, 12, 10), lo=-1.0, hi=1.0)
        yb = blk(xb)
        yb.backward(go)
        o = {"y": _np(yb), "dx": _np(xb.grad)}
        for k, p in blk.named_parameters():
            o["g." + k] = _np(p.grad)
        np.savez_compressed(os.path.join(OUT, f"promptir_block_{lnt}.npz"), **o)
    # one PromptGenBlock, up- and down-sampling resize, every gradient
    pg = ref.promptir.PromptGenBlock(prompt_dim=8, prompt_len=5, prompt_size=6, lin_dim=12)
    pg.load_state_dict({k: keyed_tensor("pg." + k, tuple(v.shape)) for k, v in pg.state_dict().items()}, strict=True)
    o = {}
    for tag, hw in (("up", (13, 9)), ("down", (4, 5)), ("same", (6, 6))):
        xp = keyed_input(f"pg.{tag}.x", (3, 12) + hw, lo=-1.0, hi=1.0).requires_grad_(True)
        go = keyed_input(f"pg.{tag}.go", (3, 8) + hw, lo=-1.0, hi=1.0)
        pg.zero_grad()
        yp = pg(xp)
        yp.backward(go)
        o[f"{tag}.y"], o[f"{tag}.dx"] = _np(yp), _np(xp.grad)
        for k, p in pg.named_parameters():
            o[f"{tag}.g.{k}"] = _np(p.grad)
    np.savez_compressed(os.path.join(OUT, "promptir_promptgen.npz"), **o)


def gen_train_plumbing(ref):
    """learning-rate sequences of the reference's two schedulers and the index maps of its ConcatDataset / EnlargedSampler
    (host-side training plumbing; basicsr/models/lr_scheduler.py, basicsr/data/concat_dataset.py, data_sampler.py)"""
    import importlib.util

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_import.REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    lrs = load("_ref_lr_scheduler", "basicsr/models/lr_scheduler.py")
    cds = load("_ref_concat_dataset", "basicsr/data/concat_dataset.py")
    smp = load("_ref_data_sampler", "basicsr/data/data_sampler.py")
    out = {}

    def run(make, n):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=2e-4)
        sch = make(opt)
        seq = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            opt.step()
            sch.step()
            seq.append(opt.param_groups[0]["lr"])
        return np.array(seq, dtype=np.float64)

    out["multistep"] = run(lambda o: lrs.MultiStepRestartLR(o, milestones=[5, 9, 9, 14], gamma=0.5), 20)
    out["multistep_restart"] = run(lambda o: lrs.MultiStepRestartLR(o, milestones=[3, 6, 13, 16], gamma=0.5, restarts=[0, 10],
                                                                   restart_weights=[1, 0.5]), 20)
    out["cosine"] = run(lambda o: lrs.CosineAnnealingRestartLR(o, periods=[8, 6, 6], restart_weights=[1, 0.5, 0.25],
                                                              eta_min=[1e-7, 2e-7, 3e-7]), 20)
    out["cosine_single"] = run(lambda o: lrs.CosineAnnealingRestartLR(o, periods=[20], restart_weights=[1], eta_min=1e-6), 20)

    class Toy(torch.utils.data.Dataset):
        def __init__(self, n, tag):
            self.n, self.tag = n, tag

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return {"v": self.tag * 100 + i}

    cat = cds.ConcatDataset([Toy(3, 1), Toy(5, 2), Toy(2, 3)], [2, 1, 3])
    out["concat_len"] = np.array(len(cat))
    out["concat_map"] = np.array([[cat[i]["v"], cat[i]["dataset_idx"]] for i in range(len(cat))] + [[cat[-1]["v"], cat[-1]["dataset_idx"]]])
    for world, rank, ratio in ((1, 0, 1), (2, 1, 3), (4, 2, 2)):
        sp = smp.EnlargedSampler(Toy(11, 0), world, rank, ratio)
        seqs = []
        for ep in (0, 3):
            sp.set_epoch(ep)
            seqs.append(list(iter(sp)))
        out[f"sampler_{world}_{rank}_{ratio}"] = np.array(seqs)
    np.savez_compressed(os.path.join(OUT, "train_plumbing.npz"), **out)


def gen_metrics(ref):
    """PSNR of the reference's own ``calculate_psnr`` (basicsr/metrics/psnr_ssim.py:11-75) on keyed float BCHW batches.
    The module imports cv2 (absent here); PSNR only uses ``cv2.cvtColor(img, COLOR_RGB2BGR)``, a channel flip, so a module
    object providing just that is put in sys.modules for the import.  ``calculate_ssim`` needs cv2.filter2D / getGaussianKernel
    and is NOT generated (SSIM stays pinned by an independent scipy implementation in tests/test_plumbing_cpu.py)."""
    import importlib.util
    import types

    saved = {k: v for k, v in sys.modules.items() if k == "basicsr" or k.startswith("basicsr.") or k == "cv2"}
    for k in saved:
        del sys.modules[k]
    try:
        cv2 = types.ModuleType("cv2")
        cv2.COLOR_RGB2BGR = 4
        cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
        sys.modules["cv2"] = cv2
        for name, path in [("basicsr", "basicsr"), ("basicsr.utils", "basicsr/utils"), ("basicsr.metrics", "basicsr/metrics")]:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(ref_import.REF, path)]
            sys.modules[name] = m

        def load(name, rel):
            spec = importlib.util.spec_from_file_location(name, os.path.join(ref_import.REF, rel))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            return mod

        load("basicsr.utils.registry", "basicsr/utils/registry.py")
        cu = load("basicsr.utils.color_util", "basicsr/utils/color_util.py")
        sys.modules["basicsr.utils"].bgr2ycbcr = cu.bgr2ycbcr
        load("basicsr.metrics.metric_util", "basicsr/metrics/metric_util.py")
        ps = load("basicsr.metrics.psnr_ssim", "basicsr/metrics/psnr_ssim.py")
        a = keyed_input("metrics.a", (2, 3, 24, 20)).numpy()
        b = np.clip(a + keyed_input("metrics.n", (2, 3, 24, 20), lo=-0.08, hi=0.08).numpy(), 0, 1).astype(np.float32)
        out = {}
        for cb in (0, 3):
            for ych in (False, True):
                out[f"psnr_cb{cb}_y{int(ych)}"] = np.float64(ps.calculate_psnr(a, b, cb, test_y_channel=ych, image_range=255.0))
        out["psnr_range1"] = np.float64(ps.calculate_psnr(a, b, 0, image_range=1))
        out["psnr_single_chw"] = np.float64(ps.calculate_psnr(a[0], b[0], 2, image_range=255.0))
        out["psnr_bhwc"] = np.float64(ps.calculate_psnr(a.transpose(0, 2, 3, 1), b.transpose(0, 2, 3, 1), 0, input_order="BHWC"))
        out["psnr_equal"] = np.float64(ps.calculate_psnr(a, a.copy(), 0))
        np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    finally:
        for k in [k for k in sys.modules if k == "basicsr" or k.startswith("basicsr.") or k == "cv2"]:
            del sys.modules[k]
        sys.modules.update(saved)


def gen_denoise_noise(ref):
    """The noise field of ``PairedImageDenoiseDataset.__getitem__`` (basicsr/data/paired_image_dataset.py:388-402).  The class
    itself needs cv2 / the file clients, so its five numpy lines are re-enacted verbatim on a keyed HWC float32 image:
    ``np.random.seed(index | 0)``; ``img_lq += np.random.normal(0, sigma / 255.0, img_lq.shape)``."""
    img = keyed_input("denoise.img", (6, 5, 3)).numpy()   # HWC, RGB order (after the reference's BGR2RGB)
    img = (np.round(img * 255.0) / 255.0).astype(np.float32)   # values a PNG can hold
    out = {"img_hwc_u8": np.round(img * 255.0).astype(np.uint8)}
    for tag, seed in (("test", 0), ("train_idx3", 3)):
        lq = img.copy()
        np.random.seed(seed=seed)
        lq += np.random.normal(0, 25 / 255.0, lq.shape)
        out[f"lq_{tag}"] = lq
    np.savez_compressed(os.path.join(OUT, "denoise_noise.npz"), **out)


def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = ref_import.load_reference_archs()
    import warnings

    warnings.filterwarnings("ignore")
    gen_ln(ref)
    gen_nafblock(ref)
    gen_nafnet_tiny(ref)
    gen_nafnet_full(ref)
    gen_tlsc(ref)
    gen_dc_head(ref)
    gen_dc_img_head(ref)
    gen_dcpt_step(ref)
    gen_dcdist_step(ref)
    gen_restormer(ref)
    gen_promptir(ref)
    gen_train_plumbing(ref)
    gen_metrics(ref)
    gen_denoise_noise(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
