"""GPU parity: degradation-classifier head (PromptIR_NoImg_DC) and its building blocks vs the oracle and
the golden vectors of the real reference."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict, keyed_tensor
from oracle import dc_oracle as D
from kernel_trace import kernel_trace

pytestmark = pytest.mark.gpu
DC_CFG = dict(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from dcpt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def relerr(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def check(name, a, b, tol):
    e = relerr(a, b)
    assert np.isfinite(e) and e <= tol, f"{name}: scale-relative max error {e:.3e} > {tol:.1e}"


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,relu,res", [
    (2, 8, 16, 6, 10, 1, True, False), (1, 16, 16, 9, 7, 3, True, False), (2, 16, 8, 8, 8, 1, True, True),
    (1, 128, 128, 16, 16, 3, True, False), (2, 64, 32, 5, 5, 1, False, False), (1, 24, 40, 4, 6, 3, True, True)])
def test_conv_ln(dev, B, Cin, Cout, H, W, ks, relu, res):
    from dcpt_amd import functional as DF

    x = keyed_input("cl.x", (B, Cin, H, W), lo=-1, hi=1)
    w = keyed_tensor("cl.conv.weight", (Cout, Cin, ks, ks))
    lw = keyed_tensor("cl.norm.weight", (Cout,))
    lb = keyed_tensor("cl.norm.bias", (Cout,))
    r = keyed_input("cl.res", (B, Cout, H, W), lo=-1, hi=1) if res else None
    go = keyed_input("cl.go", (B, Cout, H, W), lo=-1, hi=1)
    ts = [x, w, lw, lb] + ([r] if res else [])
    ref = [t.clone().requires_grad_(True) for t in ts]
    y = D.layernorm_cf(F.conv2d(ref[0], ref[1], padding=ks // 2), ref[2], ref[3])
    if res:
        y = y + ref[4]
    if relu:
        y = F.relu(y)
    y.backward(go)
    gpu = [t.to(dev).requires_grad_(True) for t in ts]
    yg = DF.conv_ln(gpu[0], gpu[1], gpu[2], gpu[3], gpu[4] if res else None, relu)
    yg.backward(go.to(dev))
    check("y", yg, y, 2e-5)
    for n, a, b in zip(["dx", "dw", "dlnw", "dlnb", "dres"], gpu, ref):
        check(n, a.grad, b.grad, 1e-4)


@pytest.mark.parametrize("B,C,H,W,bf", [(2, 16, 6, 10, False), (1, 64, 16, 16, False), (2, 16, 6, 10, True), (1, 128, 16, 16, True),
                                        (3, 64, 13, 11, True), (1, 128, 224, 224, True), (1, 40, 9, 9, True), (2, 64, 225, 223, True)])
def test_bottleneck_node(dev, B, C, H, W, bf):
    """DF.bottleneck (the BottleneckBlock of the reference, degrad_classify_arch.py:132-243, as ONE autograd node with the shortcut
    gradient summed in conv1's data-gradient GEMM: dcpt_conv_ln_bwd_acc*) against the PyTorch-CPU restatement of the reference lines (fp32)
    and against the three-node chain of conv_ln calls it replaces (both dtypes)."""
    from dcpt_amd import functional as DF

    Cb = 2 * C
    x = keyed_input("bn.x", (B, C, H, W), lo=-1, hi=1)
    ws = [keyed_tensor("bn.conv1.weight", (Cb, C, 1, 1)), keyed_tensor("bn.norm1.weight", (Cb,)), keyed_tensor("bn.norm1.bias", (Cb,)),
          keyed_tensor("bn.conv2.weight", (Cb, Cb, 3, 3)), keyed_tensor("bn.norm2.weight", (Cb,)), keyed_tensor("bn.norm2.bias", (Cb,)),
          keyed_tensor("bn.conv3.weight", (C, Cb, 1, 1)), keyed_tensor("bn.norm3.weight", (C,)), keyed_tensor("bn.norm3.bias", (C,))]
    go = keyed_input("bn.go", (B, C, H, W), lo=-1, hi=1)
    cast = (lambda t: t.to(dev).bfloat16()) if bf else (lambda t: t.to(dev))
    # one node
    xg = cast(x).requires_grad_(True)
    pg = [t.to(dev).requires_grad_(True) for t in ws]
    with kernel_trace() as tr:
        y = DF.bottleneck(xg, *pg)
        y.backward(cast(go))
    if bf:
        # the bf16 node is ONE library call per direction (dcpt_bottleneck_*_bf16): LayerNorms in the GEMM epilogues where a row fits a
        # column tile -- 2C <= 128 on the 128-row kernel; 2C == 256 on the 256-row kernel, which needs >= 192 tiles of 256 pixels
        M = B * H * W
        wide_ok = 2 * C <= 128 or (2 * C == 256 and (M + 255) // 256 >= 192)
        if wide_ok:
            tr.assert_ran("head.conv1x1+ln_fwd_epilogue", "head.conv3x3+ln_fwd_epilogue", "head.conv1x1_dgrad+ln_bwd_epilogue",
                          "head.conv3x3_dgrad+ln_bwd_epilogue")
            tr.assert_not_ran("head.conv1x1,ln_fwd_kernel", "head.conv3x3,ln_fwd_kernel")
            assert tr["head.ln_bwd_kernel"] == 1   # (the block's last LayerNorm: its gradient comes from outside)
            if 2 * C == 128 and M >= 192 * 512:   # 128 channels on the 512 x 128 tile of the 256-row kernel (stage 0 of the DCPT head)
                tr.assert_ran("nt_bf16.tall512_conv3")
                assert tr["nt_bf16.tall512_conv3"] == 1   # conv2's forward (its data gradient, with the LayerNorm-backward epilogue: 128-row kernel)
        elif 2 * C > 256:
            tr.assert_not_ran("head.conv3x3+ln_fwd_epilogue", "head.conv3x3_dgrad+ln_bwd_epilogue")
            assert tr["head.ln_bwd_kernel"] == 3
    # three nodes (autograd sums the two gradients of x itself)
    conv_ln = DF.conv_ln_bf16 if bf else DF.conv_ln
    xc = cast(x).requires_grad_(True)
    pc = [t.to(dev).requires_grad_(True) for t in ws]
    o = conv_ln(xc, pc[0], pc[1], pc[2], None, True)
    o = conv_ln(o, pc[3], pc[4], pc[5], None, True)
    yc = conv_ln(o, pc[6], pc[7], pc[8], xc, True)
    yc.backward(cast(go))
    assert torch.equal(y, yc), "the forward of the node is the three calls of the chain"
    tol = 2e-2 if bf else 1e-6   # (bf16: the chain rounds dx twice -- conv1's dx, then the sum; the node rounds the fp32 sum once)
    check("dx vs chain", xg.grad.float(), xc.grad.float(), tol)
    for n, a, b in zip(["dw1", "dlw1", "dlb1", "dw2", "dlw2", "dlb2", "dw3", "dlw3", "dlb3"], pg, pc):
        if bf and n in ("dw1", "dlw1", "dlb1", "dw2", "dlw2", "dlb2"):
            # (everything behind an inner LayerNorm's backward: in the node it runs in a GEMM epilogue, in the chain as a kernel of its own --
            # the same formulas on the same bf16 inputs, but two compilations of them (fma contraction) and two orders of the parameter
            # sums: dz differs by an ulp of bf16 in a few elements, the sums over all pixels by ~1e-4 of their scale)
            check(n, a.grad, b.grad, 1e-3)
        else:
            assert torch.equal(a.grad, b.grad), (f"{n}: the parameter gradients do not depend on where the shortcut gradient is summed "
                                                 f"(max diff {float((a.grad - b.grad).abs().max()):.3e} of {float(b.grad.abs().max()):.3e})")
    if not bf:   # the reference's lines on the CPU
        ref = [t.clone().requires_grad_(True) for t in [x] + ws]
        r = F.relu(D.layernorm_cf(F.conv2d(ref[0], ref[1]), ref[2], ref[3]))
        r = F.relu(D.layernorm_cf(F.conv2d(r, ref[4], padding=1), ref[5], ref[6]))
        r = F.relu(D.layernorm_cf(F.conv2d(r, ref[7]), ref[8], ref[9]) + ref[0])
        r.backward(go)
        check("y", y, r, 2e-5)
        check("dx", xg.grad, ref[0].grad, 1e-4)
        for n, a, b in zip(["dw1", "dlw1", "dlb1", "dw2", "dlw2", "dlb2", "dw3", "dlw3", "dlb3"], pg, ref[1:]):
            check(n, a.grad, b.grad, 1e-4)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 8, 16, 6, 10), (1, 64, 128, 16, 16), (3, 16, 16, 2, 2)])
def test_conv_pool_relu(dev, B, Cin, Cout, H, W):
    from dcpt_amd import functional as DF

    x = keyed_input("cp.x", (B, Cin, H, W), lo=-1, hi=1)
    w = keyed_tensor("cp.conv.weight", (Cout, Cin, 1, 1))
    go = keyed_input("cp.go", (B, Cout, H // 2, W // 2), lo=-1, hi=1)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.relu(F.max_pool2d(F.conv2d(xr, wr), 2, 2))
    yr.backward(go)
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = DF.conv1x1_pool_relu(xg, wg)
    y.backward(go.to(dev))
    check("y", y, yr, 1e-5)
    check("dx", xg.grad, xr.grad, 1e-4)
    check("dw", wg.grad, wr.grad, 1e-4)


def test_mix_and_head(dev):
    from dcpt_amd import functional as DF

    prev = keyed_input("mx.p", (2, 16, 6, 6), lo=-1, hi=1)
    feat = keyed_input("mx.f", (2, 16, 6, 6), lo=-1, hi=1)
    mw = keyed_tensor("mx.mixing_weights", (4,))
    go = keyed_input("mx.go", (2, 16, 6, 6), lo=-1, hi=1)
    for use_prev in (True, False):
        pr, fr, mr = prev.clone().requires_grad_(True), feat.clone().requires_grad_(True), mw.clone().requires_grad_(True)
        yr = (pr if use_prev else 0) + torch.softmax(mr, 0)[2] * fr
        yr.backward(go)
        pg, fg, mg = (t.to(dev).requires_grad_(True) for t in (prev, feat, mw))
        y = DF.mix(pg if use_prev else None, fg, mg, 2)
        y.backward(go.to(dev))
        check("mix y", y, yr, 1e-6)
        check("mix df", fg.grad, fr.grad, 1e-6)
        check("mix dw", mg.grad, mr.grad, 1e-5)
        if use_prev:
            check("mix dprev", pg.grad, pr.grad, 1e-7)
    x = keyed_input("hd.x", (3, 64, 5, 7), lo=-1, hi=1)
    fw, fb = keyed_tensor("hd.fc.weight", (10, 64)), keyed_tensor("hd.fc.bias", (10,))
    gl = keyed_input("hd.go", (3, 10), lo=-1, hi=1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, fw, fb))
    lr = F.linear(xr.mean(dim=[-1, -2]), wr, br)
    lr.backward(gl)
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, fw, fb))
    lg = DF.meanpool_fc(xg, wg, bg)
    lg.backward(gl.to(dev))
    check("logits", lg, lr, 1e-5)
    check("dx", xg.grad, xr.grad, 1e-5)
    check("dfw", wg.grad, wr.grad, 1e-5)
    check("dfb", bg.grad, br.grad, 1e-5)


def test_dc_head_golden(dev, golden_dir):
    from basicsr.archs import build_network

    g = np.load(os.path.join(golden_dir, "dc_head.npz"))
    net = build_network(dict(type="PromptIR_NoImg_DC", **DC_CFG))
    sd = keyed_state_dict(D.dc_param_shapes(**DC_CFG), seed=0)
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    feats = [keyed_input(f"dc.f{i}", (3, c, 32 >> i, 32 >> i), lo=-1.0, hi=1.0).to(dev).requires_grad_(True)
             for i, c in enumerate(DC_CFG["feature_dims"])]
    logits = net(None, feats)
    loss = F.cross_entropy(logits, torch.tensor([1, 7, 4], device=dev))
    loss.backward()
    check("logits", logits, g["logits"], 1e-4)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for i, f in enumerate(feats):
        check(f"df{i}", f.grad, g[f"df{i}"], 1e-3)
    params = dict(net.named_parameters())
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(params[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 1e-3 * max(1e-9, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            check("grad " + k[2:], params[k[2:]].grad, g[k], 1e-3)


def test_dc_img_head_golden(dev, golden_dir):
    """PromptIR_DC (reference :480-555): embedding (patch rows -> MFMA GEMM + LayerNorm) + stages vs the real reference"""
    from basicsr.archs import build_network
    from dcpt_amd import functional as DF

    g = np.load(os.path.join(golden_dir, "dc_img_head.npz"))
    net = build_network(dict(type="PromptIR_DC", **DC_CFG))
    sd = keyed_state_dict(D.dc_param_shapes(**DC_CFG, img_embed=True), seed=0)
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    lq2 = keyed_input("dci.lq2", (2, 3, 64, 32)).to(dev).requires_grad_(True)
    feats = [keyed_input(f"dci.g{i}", (2, c, 32 >> i, 16 >> i), lo=-1.0, hi=1.0).to(dev).requires_grad_(True)
             for i, c in enumerate(DC_CFG["feature_dims"])]
    logits = net(lq2, feats)
    loss = F.cross_entropy(logits, torch.tensor([5, 2], device=dev))
    loss.backward()
    check("logits", logits, g["logits"], 1e-4)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    check("dlq", lq2.grad, g["dlq"], 1e-3)
    for i, f in enumerate(feats):
        check(f"df{i}", f.grad, g[f"df{i}"], 1e-3)
    params = dict(net.named_parameters())
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(params[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 1e-3 * max(1e-9, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            check("grad " + k[2:], params[k[2:]].grad, g[k], 1e-3)
    # the embedding alone on odd-sized images (36 x 28 -> 18 x 14, 37 x 29 -> 19 x 15), with its image gradient
    conv, norm = net.conv_embed[0], net.conv_embed[1]
    lq = keyed_input("dci.lq", (3, 3, 36, 28)).to(dev).requires_grad_(True)
    e = DF.conv_embed_ln(lq, conv.weight, conv.bias, norm.weight, norm.bias)
    check("embed", e, g["embed"], 2e-5)
    (dlq,) = torch.autograd.grad((e * keyed_input("dci.ge", tuple(e.shape), lo=-1.0, hi=1.0).to(dev)).sum(), lq)
    check("embed dlq", dlq, g["embed_dlq"], 1e-4)
    with torch.no_grad():
        check("embed odd", DF.conv_embed_ln(keyed_input("dci.lq3", (1, 3, 37, 29)).to(dev), conv.weight, conv.bias, norm.weight,
                                            norm.bias), g["embed_odd"], 2e-5)
    # like the reference, feature maps at the image resolution (NAFNet's taps) are rejected
    assert bool(g["full_res_features_fail"])
    with pytest.raises(RuntimeError):
        net(lq2.detach(), [keyed_input("dci.bad", (2, 8, 64, 32)).to(dev)] + [f.detach() for f in feats[1:]])
