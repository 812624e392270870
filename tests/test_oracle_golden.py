"""CPU: pin the oracle restatement (oracle/nafnet_oracle.py) against golden vectors produced
by the real reference (oracle/make_golden.py; fixtures in tests/golden/)."""
import os

import numpy as np
import pytest
import torch

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict, keyed_tensor
from oracle import nafnet_oracle as O

TINY = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])
FULL = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])


def _close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _req(P):
    return {k: v.clone().requires_grad_(True) for k, v in P.items()}


@pytest.mark.parametrize("tag,shape", [("a", (2, 64, 8, 8)), ("b", (1, 512, 4, 4)), ("c", (3, 8, 5, 7))])
def test_ln2d(golden_dir, tag, shape):
    g = np.load(os.path.join(golden_dir, "ln2d.npz"))
    C = shape[1]
    x = keyed_input(f"ln.{tag}.x", shape, lo=-2.0, hi=3.0).requires_grad_(True)
    w = keyed_tensor(f"ln.{tag}.norm.weight", (C,)).requires_grad_(True)
    b = keyed_tensor(f"ln.{tag}.norm.bias", (C,)).requires_grad_(True)
    go = keyed_input(f"ln.{tag}.go", shape, lo=-1.0, hi=1.0)
    y = O.layernorm2d(x, w, b)
    y.backward(go)
    _close(y, g[f"{tag}.y"])
    _close(x.grad, g[f"{tag}.dx"], atol=2e-6)
    _close(w.grad, g[f"{tag}.dw"], rtol=1e-4, atol=1e-5)
    _close(b.grad, g[f"{tag}.db"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("c", [16, 64])
def test_nafblock(golden_dir, c):
    g = np.load(os.path.join(golden_dir, f"nafblock_c{c}.npz"))
    shapes = {k[len("b."):]: s for k, s in O.nafnet_param_shapes(width=c, enc_blk_nums=[1], middle_blk_num=0).items()
              if False}
    # block parameter shapes: take them from a 1-block net description
    full = O.nafnet_param_shapes(width=c, enc_blk_nums=[1], middle_blk_num=0, dec_blk_nums=[])
    P = {}
    for k, s in full.items():
        if k.startswith("encoders.0.0."):
            leaf = k[len("encoders.0.0."):]
            P[leaf] = keyed_tensor(f"blk{c}." + leaf, s)
    P = _req(P)
    x = keyed_input(f"blk{c}.x", (2, c, 16, 16), lo=-1.0, hi=1.0).requires_grad_(True)
    go = keyed_input(f"blk{c}.go", (2, c, 16, 16), lo=-1.0, hi=1.0)
    y = O.nafblock(x, P, "")
    y.backward(go)
    _close(y, g["y"], rtol=1e-5, atol=1e-5)
    _close(x.grad, g["dx"], rtol=1e-4, atol=1e-5)
    for k, p in P.items():
        _close(p.grad, g["g." + k], rtol=1e-4, atol=2e-5)


def test_nafnet_tiny(golden_dir):
    g = np.load(os.path.join(golden_dir, "nafnet_tiny.npz"))
    P = _req(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0))
    x = keyed_input("tiny.x", (2, 3, 32, 32)).requires_grad_(True)
    gw = keyed_input("tiny.gw", (2, 3, 32, 32), lo=-1.0, hi=1.0)
    y, taps = O.nafnet_forward(x, P)
    (y * gw).sum().backward()
    _close(y, g["y"], rtol=1e-4, atol=1e-5)
    for i, t in enumerate(taps):
        _close(t, g[f"tap{i}"], rtol=1e-4, atol=1e-5)
    _close(x.grad, g["dx"], rtol=1e-3, atol=1e-4)
    names = [str(n) for n in g["g_names"]]
    assert names == list(P.keys())  # same keys, same registration order as the reference
    for n, l2 in zip(names, g["g_l2"]):
        mine = float(P[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 1e-4 * max(1.0, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            ref = g[k]
            err = np.abs(P[k[2:]].grad.numpy() - ref).max()
            assert err <= 1e-5 * max(1.0, np.abs(ref).max()), (k, err)
    out, taps2 = O.nafnet_forward(x.detach(), P, hook=True)
    assert out is None and len(taps2) == 4


def test_nafnet_full_keys_and_output(golden_dir):
    g = np.load(os.path.join(golden_dir, "nafnet_full.npz"))
    shapes = O.nafnet_param_shapes(**FULL)
    assert len(shapes) == int(g["n_keys"]) == 664
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"])
    P = keyed_state_dict(shapes, seed=0)
    x = keyed_input("full.x", (1, 3, 256, 256))
    with torch.no_grad():
        y, _ = O.nafnet_forward(x, P)
    _close(y[..., ::16, ::16], g["y_sub"], rtol=1e-3, atol=1e-4)
    assert abs(float(y.double().mean()) - float(g["y_mean"])) < 1e-5
    gt = keyed_input("full.gt", (1, 3, 256, 256))
    assert abs(float(O.l1_loss(y, gt)) - float(g["loss"])) < 1e-5


def test_tlsc(golden_dir):
    g = np.load(os.path.join(golden_dir, "tlsc.npz"))
    x = keyed_input("tlsc.x", (2, 8, 48, 40), lo=-1.0, hi=1.0)
    y = O.tlsc_avgpool(x, tuple(int(k) for k in g["kernel"]))
    _close(y, g["y"], rtol=1e-5, atol=1e-6)


def test_dc_head(golden_dir):
    from oracle import dc_oracle as D

    g = np.load(os.path.join(golden_dir, "dc_head.npz"))
    cfg = dict(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10)
    shapes = D.dc_param_shapes(**cfg)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]] and len(shapes) == 97
    P = _req(keyed_state_dict(shapes, seed=0))
    feats = [keyed_input(f"dc.f{i}", (3, c, 32 >> i, 32 >> i), lo=-1.0, hi=1.0).requires_grad_(True)
             for i, c in enumerate(cfg["feature_dims"])]
    logits = D.dc_forward(feats, P)
    loss = torch.nn.functional.cross_entropy(logits, torch.tensor([1, 7, 4]))
    loss.backward()
    _close(logits, g["logits"], rtol=1e-4, atol=1e-5)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for i, f in enumerate(feats):
        ref = g[f"df{i}"]
        assert np.abs(f.grad.numpy() - ref).max() <= 1e-4 * max(1e-6, np.abs(ref).max()), i
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(P[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 1e-4 * max(1e-6, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            ref = g[k]
            assert np.abs(P[k[2:]].grad.numpy() - ref).max() <= 1e-4 * max(1e-6, np.abs(ref).max()), k


def test_dc_img_head(golden_dir):
    """PromptIR_DC (7x7 stride-2 image embedding + the same stages) == the reference: logits, loss, image / feature /
    parameter gradients, the embedding on odd-sized images, and the reference's refusal of full-resolution features"""
    from oracle import dc_oracle as D

    g = np.load(os.path.join(golden_dir, "dc_img_head.npz"))
    cfg = dict(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10)
    shapes = D.dc_param_shapes(**cfg, img_embed=True)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]] and len(shapes) == 101
    P = _req(keyed_state_dict(shapes, seed=0))
    lq2 = keyed_input("dci.lq2", (2, 3, 64, 32)).requires_grad_(True)
    feats = [keyed_input(f"dci.g{i}", (2, c, 32 >> i, 16 >> i), lo=-1.0, hi=1.0).requires_grad_(True)
             for i, c in enumerate(cfg["feature_dims"])]
    logits = D.dc_img_forward(lq2, feats, P)
    loss = torch.nn.functional.cross_entropy(logits, torch.tensor([5, 2]))
    loss.backward()
    _close(logits, g["logits"], rtol=1e-4, atol=1e-5)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for name, t in [("dlq", lq2)] + [(f"df{i}", f) for i, f in enumerate(feats)]:
        ref = g[name]
        assert np.abs(t.grad.numpy() - ref).max() <= 1e-4 * max(1e-6, np.abs(ref).max()), name
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(P[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 1e-4 * max(1e-6, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            ref = g[k]
            assert np.abs(P[k[2:]].grad.numpy() - ref).max() <= 1e-4 * max(1e-6, np.abs(ref).max()), k
    with torch.no_grad():
        embed = lambda x: D.layernorm_cf(torch.nn.functional.conv2d(x, P["conv_embed.0.weight"], P["conv_embed.0.bias"], stride=2,
                                                                    padding=3), P["conv_embed.1.weight"], P["conv_embed.1.bias"])
        _close(embed(keyed_input("dci.lq", (3, 3, 36, 28))), g["embed"], rtol=1e-4, atol=1e-5)
        _close(embed(keyed_input("dci.lq3", (1, 3, 37, 29))), g["embed_odd"], rtol=1e-4, atol=1e-5)
    assert bool(g["full_res_features_fail"])
    with pytest.raises(RuntimeError):
        D.dc_img_forward(lq2.detach(), [keyed_input("dci.bad", (2, 8, 64, 32))] + [f.detach() for f in feats[1:]], P)


def test_dcpt_step(golden_dir):
    """oracle re-enactment of DCPTModel.optimize_parameters == the reference's (losses + all grad norms)."""
    from oracle import dc_oracle as D

    g = np.load(os.path.join(golden_dir, "dcpt_step.npz"))
    Pg = _req(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0))
    Pd = _req(keyed_state_dict(D.dc_param_shapes(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10), seed=0))
    gt = keyed_input("dcpt.gt", (2, 3, 32, 32))
    lq = keyed_input("dcpt.lq", (2, 3, 32, 32))
    pix, _ = O.nafnet_forward(gt, Pg)
    l_pix = O.l1_loss(pix, gt)
    none, taps = O.nafnet_forward(lq, Pg, hook=True)
    assert none is None
    logits = D.dc_forward(taps[::-1], Pd)
    l_cls = torch.nn.functional.cross_entropy(logits, torch.tensor([3, 8]))
    (l_pix + l_cls).backward()
    assert abs(float(l_pix) - float(g["l_pix"])) < 1e-6 and abs(float(l_cls) - float(g["l_classify"])) < 1e-5
    for tag, P in (("g", Pg), ("dc", Pd)):
        for n, l2 in zip([str(s) for s in g[f"{tag}_names"]], g[f"{tag}_l2"]):
            mine = float(P[n].grad.double().pow(2).sum().sqrt())
            assert abs(mine - l2) <= 2e-4 * max(1e-7, l2), (tag, n, mine, l2)


def test_dcdist_step(golden_dir):
    """oracle re-enactment of DCDistModel.optimize_parameters (one net_g forward with taps on the last block of each decoder
    level, frozen head, L1 + CE) == the reference's: losses, logits, restored image, all net_g gradient norms"""
    from basicsr.archs import build_network
    from oracle import dc_oracle as D
    from oracle import restormer_oracle as R

    g = np.load(os.path.join(golden_dir, "dcdist_step.npz"))
    cfg_g = dict(dim=16, num_blocks=[4, 6, 6, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8])
    cfg_dc = dict(feature_dims=[32, 32, 64], num_res_blocks=1, num_classes=5)
    shapes = {k: tuple(v.shape) for k, v in build_network(dict(type="Restormer_origin", **cfg_g)).state_dict().items()}
    Pg = _req(keyed_state_dict(shapes, seed=0))
    Pd = keyed_state_dict(D.dc_param_shapes(**cfg_dc), seed=0)   # frozen
    lq, gt = keyed_input("dist.lq", (2, 3, 32, 32)), keyed_input("dist.gt", (2, 3, 32, 32))
    pix, taps = R.restormer_forward(lq, Pg, origin=True)
    assert [list(t.shape) for t in taps] == g["tap_shapes"].tolist()
    logits = D.dc_forward(taps[::-1], Pd)
    l_pix = (pix - gt).abs().mean()
    l_cls = torch.nn.functional.cross_entropy(logits, torch.tensor([4, 1]))
    (l_pix + l_cls).backward()
    assert abs(float(l_pix) - float(g["l_pixel"])) < 1e-5 and abs(float(l_cls) - float(g["l_classify"])) < 1e-5
    _close(logits, g["logits"], rtol=1e-4, atol=1e-5)
    _close(pix, g["pix"], rtol=1e-4, atol=1e-5)
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(Pg[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 2e-4 * max(1e-7, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            ref = g[k]
            assert np.abs(Pg[k[2:]].grad.numpy() - ref).max() <= 2e-4 * max(1e-7, np.abs(ref).max()), k


R_CFG = dict(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8])


def _restormer_shapes(name):
    """state-dict shapes of the product arch (checked key-by-key against the reference's own key list in the fixture)"""
    from basicsr.archs import build_network

    return {k: tuple(v.shape) for k, v in build_network(dict(type=name, **R_CFG)).state_dict().items()}


@pytest.mark.parametrize("tag,origin", [("restormer", False), ("restormer_origin", True)])
def test_restormer_tiny(golden_dir, tag, origin):
    from oracle import restormer_oracle as R

    g = np.load(os.path.join(golden_dir, f"{tag}_tiny.npz"))
    shapes = _restormer_shapes("Restormer_origin" if origin else "Restormer")
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]
    P = _req(keyed_state_dict(shapes, seed=0))
    x = keyed_input(f"{tag}.x", (2, 3, 32, 32)).requires_grad_(True)
    gw = keyed_input(f"{tag}.gw", (2, 3, 32, 32), lo=-1.0, hi=1.0)
    y, taps = R.restormer_forward(x, P, origin=origin)
    (y * gw).sum().backward()
    _close(y, g["y"], rtol=1e-4, atol=1e-5)
    assert np.abs(x.grad.numpy() - g["dx"]).max() <= 1e-4 * np.abs(g["dx"]).max()
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(P[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 2e-4 * max(1e-7, l2), (n, mine, l2)
    assert len(taps) == 3 and R.restormer_forward(x.detach(), P, hook=True, origin=origin)[0] is None


@pytest.mark.parametrize("lnt", ["BiasFree", "WithBias"])
def test_restormer_block(golden_dir, lnt):
    from oracle import restormer_oracle as R

    g = np.load(os.path.join(golden_dir, f"restormer_block_{lnt}.npz"))
    P = _req({k[2:]: keyed_tensor(f"tb{lnt}." + k[2:], g[k].shape) for k in g.files if k.startswith("g.")})
    x = keyed_input(f"tb{lnt}.x", (2, 48, 12, 10), lo=-1.0, hi=1.0).requires_grad_(True)
    go = keyed_input(f"tb{lnt}.go", (2, 48, 12, 10), lo=-1.0, hi=1.0)
    y = R.transformer_block(x, P, "")
    y.backward(go)
    _close(y, g["y"], rtol=1e-4, atol=1e-5)
    assert np.abs(x.grad.numpy() - g["dx"]).max() <= 1e-4 * np.abs(g["dx"]).max()
    for k in P:
        ref = g["g." + k]
        assert np.abs(P[k].grad.numpy() - ref).max() <= 1e-4 * max(1e-7, np.abs(ref).max()), k


# ------------------------------------------------------------------------------------------------ PromptIR
P_CFG = dict(num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
PG_SHAPES = {"prompt_param": (1, 5, 8, 6, 6), "linear_layer.weight": (5, 12), "linear_layer.bias": (5,), "conv3x3.weight": (8, 8, 3, 3)}


def test_promptir_tiny(golden_dir):
    from basicsr.archs import build_network
    from oracle import promptir_oracle as PO

    g = np.load(os.path.join(golden_dir, "promptir_tiny.npz"))
    shapes = {k: tuple(v.shape) for k, v in build_network(dict(type="PromptIR", **P_CFG)).state_dict().items()}
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]
    P = _req(keyed_state_dict(shapes, seed=0))
    x = keyed_input("pir.x", (2, 3, 64, 64)).requires_grad_(True)
    gw = keyed_input("pir.gw", (2, 3, 64, 64), lo=-1.0, hi=1.0)
    y = PO.promptir_forward(x, P)
    (y * gw).sum().backward()
    _close(y, g["y"], rtol=1e-4, atol=1e-5)
    assert np.abs(x.grad.numpy() - g["dx"]).max() <= 1e-4 * np.abs(g["dx"]).max()
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(P[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 2e-4 * max(1e-7, l2), (n, mine, l2)
    assert PO.promptir_forward(x.detach(), P, hook=True) is None
    with torch.no_grad():
        _close(PO.promptir_forward(keyed_input("pir.xs", (1, 3, 40, 24)), P), g["y_small"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("lnt", ["BiasFree", "WithBias"])
def test_promptir_block(golden_dir, lnt):
    from oracle import promptir_oracle as PO

    g = np.load(os.path.join(golden_dir, f"promptir_block_{lnt}.npz"))
    P = _req({k[2:]: keyed_tensor(f"ptb{lnt}." + k[2:], g[k].shape) for k in g.files if k.startswith("g.")})
    x = keyed_input(f"ptb{lnt}.x", (2, 48, 12, 10), lo=-1.0, hi=1.0).requires_grad_(True)
    y = PO.transformer_block(x, P, "")
    y.backward(keyed_input(f"ptb{lnt}.go", (2, 48, 12, 10), lo=-1.0, hi=1.0))
    _close(y, g["y"], rtol=1e-4, atol=1e-5)
    assert np.abs(x.grad.numpy() - g["dx"]).max() <= 1e-4 * np.abs(g["dx"]).max()
    for k in P:
        ref = g["g." + k]
        assert np.abs(P[k].grad.numpy() - ref).max() <= 1e-4 * max(1e-7, np.abs(ref).max()), k


def test_promptir_promptgen(golden_dir):
    from oracle import promptir_oracle as PO

    g = np.load(os.path.join(golden_dir, "promptir_promptgen.npz"))
    for tag, hw in (("up", (13, 9)), ("down", (4, 5)), ("same", (6, 6))):
        P = _req({k: keyed_tensor("pg." + k, s) for k, s in PG_SHAPES.items()})
        x = keyed_input(f"pg.{tag}.x", (3, 12) + hw, lo=-1.0, hi=1.0).requires_grad_(True)
        y = PO.prompt_block(x, P, "")
        y.backward(keyed_input(f"pg.{tag}.go", (3, 8) + hw, lo=-1.0, hi=1.0))
        _close(y, g[f"{tag}.y"], rtol=1e-5, atol=1e-6)
        assert np.abs(x.grad.numpy() - g[f"{tag}.dx"]).max() <= 1e-5 * max(1e-9, np.abs(g[f"{tag}.dx"]).max())
        for k in P:
            ref = g[f"{tag}.g.{k}"]
            assert np.abs(P[k].grad.numpy() - ref).max() <= 1e-5 * max(1e-9, np.abs(ref).max()), (tag, k)
