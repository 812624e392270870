"""GPU parity tests: the HIP path (through the C ABI / autograd bindings) against the oracle
(oracle/nafnet_oracle.py, CPU fp32) on the same seeded inputs and against the committed golden
vectors produced by the real reference.  Tolerances: fp32, <=1e-4 scale-relative per op, <=1e-3
end to end (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict, keyed_tensor
from oracle import nafnet_oracle as O

pytestmark = pytest.mark.gpu

TINY = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])
FULL = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dcpt_amd import _lib

    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def relerr(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def check(name, a, b, tol):
    e = relerr(a, b)
    assert np.isfinite(e) and e <= tol, f"{name}: scale-relative max error {e:.3e} > {tol:.1e}"


def block_params(c, prefix):
    full = O.nafnet_param_shapes(width=c, enc_blk_nums=[1], middle_blk_num=0, dec_blk_nums=[])
    return {k[len("encoders.0.0."):]: keyed_tensor(prefix + k[len("encoders.0.0."):], s)
            for k, s in full.items() if k.startswith("encoders.0.0.")}


FUSED = {"norm1_w": "norm1.weight", "norm1_b": "norm1.bias", "conv1_w": "conv1.weight", "conv1_b": "conv1.bias",
         "conv2_w": "conv2.weight", "conv2_b": "conv2.bias", "conv3_w": "conv3.weight", "conv3_b": "conv3.bias",
         "sca_w": "sca.1.weight", "sca_b": "sca.1.bias", "norm2_w": "norm2.weight", "norm2_b": "norm2.bias",
         "conv4_w": "conv4.weight", "conv4_b": "conv4.bias", "conv5_w": "conv5.weight", "conv5_b": "conv5.bias",
         "beta": "beta", "gamma": "gamma"}


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,shape", [("a", (2, 64, 8, 8)), ("b", (1, 512, 4, 4)), ("c", (3, 8, 5, 7))])
def test_ln2d_golden(dev, golden_dir, tag, shape):
    from dcpt_amd import functional as DF

    g = np.load(os.path.join(golden_dir, "ln2d.npz"))
    C = shape[1]
    x = keyed_input(f"ln.{tag}.x", shape, lo=-2.0, hi=3.0).to(dev).requires_grad_(True)
    w = keyed_tensor(f"ln.{tag}.norm.weight", (C,)).to(dev).requires_grad_(True)
    b = keyed_tensor(f"ln.{tag}.norm.bias", (C,)).to(dev).requires_grad_(True)
    go = keyed_input(f"ln.{tag}.go", shape, lo=-1.0, hi=1.0).to(dev)
    y = DF.layernorm2d(x, w, b, 1e-6)
    y.backward(go)
    check("y", y, g[f"{tag}.y"], 1e-5)
    check("dx", x.grad, g[f"{tag}.dx"], 1e-5)
    check("dw", w.grad, g[f"{tag}.dw"], 1e-5)
    check("db", b.grad, g[f"{tag}.db"], 1e-5)


@pytest.mark.parametrize("B,C,H,W", [(1, 1024, 3, 3), (2, 24, 6, 10), (4, 128, 16, 16)])
def test_ln2d_oracle(dev, B, C, H, W):
    from dcpt_amd import functional as DF

    x = keyed_input(f"lnx{C}", (B, C, H, W), lo=-3, hi=3)
    w = keyed_tensor(f"lnw{C}.norm.weight", (C,))
    b = keyed_tensor(f"lnb{C}.norm.bias", (C,))
    go = keyed_input(f"lngo{C}", (B, C, H, W), lo=-1, hi=1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    O.layernorm2d(xr, wr, br).backward(go)
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = DF.layernorm2d(xg, wg, bg, 1e-6)
    y.backward(go.to(dev))
    check("y", y, O.layernorm2d(x, w, b), 1e-5)
    check("dx", xg.grad, xr.grad, 1e-5)
    check("dw", wg.grad, wr.grad, 1e-5)
    check("db", bg.grad, br.grad, 1e-5)


# ---------------------------------------------------------------------------------------------
def _run_block(dev, P, x, go):
    from dcpt_amd import functional as DF

    Pg = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
    xg = x.to(dev).requires_grad_(True)
    y = DF.nafblock(xg, {fk: Pg[rk] for fk, rk in FUSED.items()})
    y.backward(go.to(dev))
    return y, xg.grad, {k: v.grad for k, v in Pg.items()}


@pytest.mark.parametrize("c", [16, 64])
def test_nafblock_golden(dev, golden_dir, c):
    g = np.load(os.path.join(golden_dir, f"nafblock_c{c}.npz"))
    P = block_params(c, f"blk{c}.")
    x = keyed_input(f"blk{c}.x", (2, c, 16, 16), lo=-1.0, hi=1.0)
    go = keyed_input(f"blk{c}.go", (2, c, 16, 16), lo=-1.0, hi=1.0)
    y, dx, grads = _run_block(dev, P, x, go)
    check("y", y, g["y"], 2e-5)
    check("dx", dx, g["dx"], 5e-5)
    for k in P:
        check("grad " + k, grads[k], g["g." + k], 1e-4)


# the last three shapes walk the depthwise ring kernels' other tilings (dwring.hip): several column tiles with recomputed edge
# columns (W > 64), the 8-piece single tile (32 < W <= 64), and two row parts (H >= 64 with few blocks)
# (3, 64, 5, 7) / (5, 64, 48, 40): the fused forward chains of the narrowest level (ffn_f32.hip) with a ragged last group of 32 pixels
# (M = 105) and with more groups than the launch has waves' ring slots (M = 9600)
@pytest.mark.parametrize("B,c,H,W", [(1, 8, 5, 7), (3, 24, 9, 4), (2, 128, 12, 20), (1, 512, 8, 8), (2, 32, 33, 17),
                                     (1, 16, 11, 70), (1, 24, 6, 50), (1, 8, 70, 9), (3, 64, 5, 7), (5, 64, 48, 40)])
def test_nafblock_oracle(dev, B, c, H, W):
    P = block_params(c, f"ob{c}.")
    x = keyed_input(f"ob{c}.x", (B, c, H, W), lo=-1.0, hi=1.0)
    go = keyed_input(f"ob{c}.go", (B, c, H, W), lo=-1.0, hi=1.0)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.nafblock(xr, Pr, "")
    yr.backward(go)
    y, dx, grads = _run_block(dev, P, x, go)
    check("y", y, yr, 2e-5)
    check("dx", dx, xr.grad, 5e-5)
    for k in P:
        check("grad " + k, grads[k], Pr[k].grad, 1e-4)


def test_nafblock_zero_gain(dev):
    """beta = gamma = 0 (the reference's init, nafnet_arch.py:162-163): identity forward, and the
    beta/gamma gradients must still be exact (they do not vanish)."""
    c = 16
    P = block_params(c, "zg.")
    P["beta"] = torch.zeros_like(P["beta"])
    P["gamma"] = torch.zeros_like(P["gamma"])
    x = keyed_input("zg.x", (2, c, 8, 8), lo=-1, hi=1)
    go = keyed_input("zg.go", (2, c, 8, 8), lo=-1, hi=1)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    O.nafblock(xr, Pr, "").backward(go)
    y, dx, grads = _run_block(dev, P, x, go)
    assert torch.equal(y.cpu(), x), "zero beta/gamma must make the block an exact identity"
    check("dx", dx, xr.grad, 1e-5)
    check("dbeta", grads["beta"], Pr["beta"].grad, 1e-4)
    check("dgamma", grads["gamma"], Pr["gamma"].grad, 1e-4)


# ---------------------------------------------------------------------------------------------
# (Cs <= 3 with Cb = 32 / 64 take the MFMA forms of conv3x3.hip: odd widths, partial 32-pixel tiles / 30-column bands, several row strips)
@pytest.mark.parametrize("B,Cs,Cb,H,W", [(2, 3, 8, 6, 10), (1, 3, 64, 16, 16), (2, 1, 16, 5, 5), (1, 4, 32, 7, 3), (2, 3, 64, 37, 45),
                                        (1, 3, 32, 40, 33), (2, 2, 64, 70, 64), (1, 1, 32, 33, 95), (3, 3, 64, 9, 130), (1, 3, 64, 1, 1)])
def test_edge_convs(dev, B, Cs, Cb, H, W):
    from dcpt_amd import functional as DF

    x = keyed_input("ei.x", (B, Cs, H, W), lo=-1, hi=1)
    wi = keyed_tensor("ei.intro.weight", (Cb, Cs, 3, 3))
    bi = keyed_tensor("ei.intro.bias", (Cb,))
    go = keyed_input("ei.go", (B, Cb, H, W), lo=-1, hi=1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, wi, bi))
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(go)
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, wi, bi))
    y = DF.conv3x3_in(xg, wg, bg)
    y.backward(go.to(dev))
    check("intro y", y, yr, 1e-5)
    check("intro dx", xg.grad, xr.grad, 1e-5)
    check("intro dw", wg.grad, wr.grad, 1e-5)
    check("intro db", bg.grad, br.grad, 1e-5)

    f = keyed_input("eo.f", (B, Cb, H, W), lo=-1, hi=1)
    we = keyed_tensor("eo.ending.weight", (Cs, Cb, 3, 3))
    be = keyed_tensor("eo.ending.bias", (Cs,))
    res = keyed_input("eo.res", (B, Cs, H, W))
    go2 = keyed_input("eo.go", (B, Cs, H, W), lo=-1, hi=1)
    fr, wr, br, rr = (t.clone().requires_grad_(True) for t in (f, we, be, res))
    yr = F.conv2d(fr, wr, br, padding=1) + rr
    yr.backward(go2)
    fg, wg, bg, rg = (t.to(dev).requires_grad_(True) for t in (f, we, be, res))
    y = DF.conv3x3_out(fg, wg, bg, rg)
    y.backward(go2.to(dev))
    check("ending y", y, yr, 1e-5)
    check("ending dx", fg.grad, fr.grad, 1e-5)
    check("ending dw", wg.grad, wr.grad, 1e-5)
    # (Cs numbers, each a sum of B H W gradient values of both signs: with Cs = 1 the "tensor's scale" is ONE cancelling sum -- randomized
    # shapes, tests/fuzz_shapes.py seed 6, found 1.3e-5 on (1, 1, 128, 38, 31) from the summation order alone)
    check("ending db", bg.grad, br.grad, 1e-5 if Cs > 1 else 5e-5)
    check("ending dres", rg.grad, rr.grad, 1e-6)


@pytest.mark.parametrize("B,C,H,W", [(2, 8, 4, 6), (1, 64, 16, 16), (3, 24, 2, 10), (1, 256, 8, 4)])
def test_down_up(dev, B, C, H, W):
    from dcpt_amd import functional as DF

    x = keyed_input("d.x", (B, C, H, W), lo=-1, hi=1)
    w = keyed_tensor("d.downs.weight", (2 * C, C, 2, 2))
    b = keyed_tensor("d.downs.bias", (2 * C,))
    go = keyed_input("d.go", (B, 2 * C, H // 2, W // 2), lo=-1, hi=1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=2)
    yr.backward(go)
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = DF.down2x2(xg, wg, bg)
    y.backward(go.to(dev))
    check("down y", y, yr, 1e-5)
    check("down dx", xg.grad, xr.grad, 1e-5)
    check("down dw", wg.grad, wr.grad, 1e-5)
    check("down db", bg.grad, br.grad, 1e-5)

    wu = keyed_tensor("u.ups.weight", (2 * C, C, 1, 1))
    skip = keyed_input("u.skip", (B, C // 2, 2 * H, 2 * W), lo=-1, hi=1)
    go2 = keyed_input("u.go", (B, C // 2, 2 * H, 2 * W), lo=-1, hi=1)
    xr, wr, sr = (t.clone().requires_grad_(True) for t in (x, wu, skip))
    yr = O.pixel_shuffle2(F.conv2d(xr, wr)) + sr
    yr.backward(go2)
    xg, wg, sg = (t.to(dev).requires_grad_(True) for t in (x, wu, skip))
    y = DF.up_ps(xg, wg, sg)
    y.backward(go2.to(dev))
    check("up y", y, yr, 1e-5)
    check("up dx", xg.grad, xr.grad, 1e-5)
    check("up dw", wg.grad, wr.grad, 1e-5)
    check("up dskip", sg.grad, sr.grad, 1e-6)


@pytest.mark.parametrize("B,C,H,W,bf", [(2, 8, 4, 6, False), (1, 64, 16, 16, False), (2, 16, 8, 4, True), (1, 64, 16, 16, True)])
def test_down_skip_node(dev, B, C, H, W, bf):
    """DF.down2x2_skip: an encoder group's output with its two consumers (down layer, skip connection; reference nafnet_arch.py:255-258,
    :264-265) as one autograd node, the skip's gradient summed in the down layer's scatter epilogue (dcpt_down2x2_bwd_acc*): against
    F.conv2d + a second use of x on the CPU (fp32) and against the two-node form (both dtypes)."""
    from dcpt_amd import functional as DF

    x = keyed_input("ds.x", (B, C, H, W), lo=-1, hi=1)
    w = keyed_tensor("ds.downs.weight", (2 * C, C, 2, 2))
    b = keyed_tensor("ds.downs.bias", (2 * C,))
    go = keyed_input("ds.go", (B, 2 * C, H // 2, W // 2), lo=-1, hi=1)
    gs = keyed_input("ds.gs", (B, C, H, W), lo=-1, hi=1)
    cast = (lambda t: t.to(dev).bfloat16()) if bf else (lambda t: t.to(dev))
    xg, wg, bg = cast(x).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y, skip = DF.down2x2_skip(xg, wg, bg)
    assert torch.equal(skip, xg)   # (a view of x wherever x is already NHWC, as inside the network)
    torch.autograd.backward([y, skip], [cast(go), cast(gs)])
    x2, w2, b2 = cast(x).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y2 = DF.down2x2(x2, w2, b2)
    torch.autograd.backward([y2, x2 * 1.0], [cast(go), cast(gs)])
    assert torch.equal(y, y2)
    assert torch.equal(wg.grad, w2.grad) and torch.equal(bg.grad, b2.grad)
    check("dx vs two nodes", xg.grad.float(), x2.grad.float(), 2e-2 if bf else 1e-6)
    if not bf:
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        yr = F.conv2d(xr, wr, br, stride=2)
        torch.autograd.backward([yr, xr * 1.0], [go, gs])
        check("y", y, yr, 1e-5)
        check("dx", xg.grad, xr.grad, 1e-5)
        check("dw", wg.grad, wr.grad, 1e-5)
    # only one of the two outputs used downstream
    x3 = cast(x).requires_grad_(True)
    y3, s3 = DF.down2x2_skip(x3, wg, bg)
    s3.backward(cast(gs))
    assert torch.equal(x3.grad, cast(gs))


def test_fused_leaky_relu(dev):
    from dcpt_amd import functional as DF

    x = keyed_input("fl.x", (2, 6, 5, 7), lo=-2, hi=2)
    b = keyed_tensor("fl.bias", (6,))
    go = keyed_input("fl.go", (2, 6, 5, 7), lo=-1, hi=1)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.leaky_relu(xr + br.view(1, -1, 1, 1), 0.2) * 2 ** 0.5  # closed form of fused_bias_act_kernel.cu:37-47
    yr.backward(go)
    xg, bg = x.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = DF.fused_leaky_relu(xg, bg)
    y.backward(go.to(dev))
    check("y", y, yr, 1e-6)
    check("dx", xg.grad, xr.grad, 1e-6)
    check("db", bg.grad, br.grad, 1e-5)


# ---------------------------------------------------------------------------------------------
def _build_net(cfg, dev):
    from basicsr.archs import build_network

    net = build_network(dict(type="NAFNetBaseline", **cfg))
    sd = keyed_state_dict(O.nafnet_param_shapes(**cfg), seed=0)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    return net.to(dev)


def test_nafnet_tiny_golden(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "nafnet_tiny.npz"))
    net = _build_net(TINY, dev)
    x = keyed_input("tiny.x", (2, 3, 32, 32)).to(dev).requires_grad_(True)
    gw = keyed_input("tiny.gw", (2, 3, 32, 32), lo=-1.0, hi=1.0).to(dev)
    taps = []
    hooks = [getattr(net, f"decoder{i}").register_forward_hook(lambda m, i, o: taps.append(o)) for i in range(4)]
    y = net(x)
    (y * gw).sum().backward()
    check("y", y, g["y"], 1e-4)
    for i, t in enumerate(taps):
        assert t.shape == g[f"tap{i}"].shape
        check(f"tap{i}", t, g[f"tap{i}"], 1e-4)
    check("dx", x.grad, g["dx"], 1e-3)
    params = dict(net.named_parameters())
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(params[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 1e-3 * max(1e-6, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            check("grad " + k[2:], params[k[2:]].grad, g[k], 1e-3)
    for h in hooks:
        h.remove()
    assert net(x.detach(), hook=True) is None


def test_nafnet_full_golden(dev, golden_dir):
    """NAFNet-64 [1,1,1,28] (options/all_in_one/test/test_NAFNet_5d.yml) on one 256x256 image."""
    g = np.load(os.path.join(golden_dir, "nafnet_full.npz"))
    net = _build_net(FULL, dev)
    assert len(net.state_dict()) == 664
    x = keyed_input("full.x", (1, 3, 256, 256)).to(dev).requires_grad_(True)
    gt = keyed_input("full.gt", (1, 3, 256, 256)).to(dev)
    y = net(x)
    loss = (y - gt).abs().mean()
    loss.backward()
    check("y_sub", y[..., ::16, ::16], g["y_sub"], 1e-3)
    assert abs(float(y.double().mean()) - float(g["y_mean"])) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    params = dict(net.named_parameters())
    bad = []
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(params[n].grad.double().pow(2).sum().sqrt())
        if abs(mine - l2) > 2e-3 * max(1e-9, l2):
            bad.append((n, mine, float(l2)))
    assert not bad, f"{len(bad)} parameter-gradient norms off by >2e-3: {bad[:8]}"
    check("dx_sub", x.grad[..., ::16, ::16], g["dx_sub"], 2e-3)


def test_batch_consistency_full_size(dev):
    """Size-independent property at the bench size (256x256): images are independent, so a batch of
    two copies gives two identical outputs equal to the single-image output (bit-exact: the kernels are
    deterministic and no reduction crosses the batch in forward)."""
    net = _build_net(FULL, dev)
    x1 = keyed_input("bc.x", (1, 3, 256, 256)).to(dev)
    with torch.no_grad():
        y1 = net(x1)
        y2 = net(torch.cat([x1, x1], 0))
    assert torch.equal(y2[0], y2[1])
    check("batch vs single", y2[0:1], y1, 1e-6)


def test_directional_derivative_full_size(dev):
    """Size-independent property at BASELINE.json's full size (NAFNet-64 [1,1,1,28], B = 32, 256 x 256): along a random direction
    d in parameter AND input space, the analytic derivative <grad L, d> of the whole backward pass equals the central difference
    (L(w + e d) - L(w - e d)) / 2e of two more forward passes.  The loss is a smooth one (mean square), reduced in fp64 so
    that only the fp32 forward noise and the O(e^2) truncation remain."""
    net = _build_net(FULL, dev)
    x = torch.rand((32, 3, 256, 256), generator=torch.Generator().manual_seed(11)).to(dev).requires_grad_(True)
    gt = torch.rand((32, 3, 256, 256), generator=torch.Generator().manual_seed(12)).to(dev)

    def loss_of(inp):
        return (net(inp).double() - gt.double()).pow(2).mean()

    loss = loss_of(x)
    loss.backward()
    gen = torch.Generator().manual_seed(13)
    params = [p for p in net.parameters()]
    dirs = [torch.randn(p.shape, generator=gen).to(dev) * p.detach().abs().mean().clamp_min(1e-3) for p in params]
    dx = torch.randn(x.shape, generator=gen).to(dev) * 0.1
    analytic = float(sum((p.grad.double() * d.double()).sum() for p, d in zip(params, dirs)) + (x.grad.double() * dx.double()).sum())
    eps = 2e-3
    vals = []
    with torch.no_grad():
        for sign in (+1.0, -1.0):
            for p, d in zip(params, dirs):
                p.add_(d, alpha=sign * eps)
            vals.append(float(loss_of(x.detach() + sign * eps * dx)))
            for p, d in zip(params, dirs):
                p.sub_(d, alpha=sign * eps)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(analytic) > 1e-4, analytic
    assert abs(numeric - analytic) <= 2e-2 * abs(analytic), (numeric, analytic, float(loss))


def test_nafnet_local_tlsc_golden(dev, golden_dir):
    """TLSC `NAFNet` (N10): local-window SCA at inference, vs the real reference's output."""
    from basicsr.archs import build_network

    g = np.load(os.path.join(golden_dir, "nafnet_local_tiny.npz"))
    net = build_network(dict(type="NAFNet", train_size=(1, 3, 16, 16), **TINY))
    ks = [tuple(b.sca[0].kernel_size) for b in net.encoders[0]] + [tuple(net.middle_blks[0].sca[0].kernel_size)]
    assert ks == [(24, 24), (1, 1)]
    net.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
    net = net.to(dev).eval()
    x = keyed_input("tlsc.img", (1, 3, 48, 32)).to(dev)
    with torch.no_grad():
        y = net(x)
    check("y", y, g["y"], 2e-4)
    with pytest.raises(NotImplementedError):
        net(x.requires_grad_(True))


def test_tiled_inference_matches_per_tile_loop(dev):
    """SRModel.test_tile (batched by tile shape) == the reference's one-tile-at-a-time loop (sr_model.py:273-361)."""
    from basicsr.models import build_model

    opt = dict(name="t", model_type="SRModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=False,
               network_g=dict(type="NAFNetBaseline", window_size=16, **TINY), path=dict(), tile=dict(infer_size=32, tile_pad=16),
               val=dict(save_img=False))
    m = build_model(opt)
    m.net_g.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
    img = keyed_input("tile.img", (1, 3, 75, 88))
    m.feed_data({"lq": img})
    m.pre_test()
    assert m.lq.shape == (1, 3, 80, 96)
    m.test_tile()
    m.post_test()
    got = m.output.cpu()
    lq = torch.nn.functional.pad(img, (0, 8, 0, 5), "reflect").to(dev)
    ref = torch.zeros_like(lq)
    with torch.no_grad():
        for y0 in range(0, 80, 32):
            for x0 in range(0, 96, 32):
                x1, y1 = min(x0 + 32, 96), min(y0 + 32, 80)
                xp0, yp0, xp1, yp1 = max(x0 - 16, 0), max(y0 - 16, 0), min(x1 + 16, 96), min(y1 + 16, 80)
                out = m.net_g(lq[:, :, yp0:yp1, xp0:xp1].contiguous())
                ref[:, :, y0:y1, x0:x1] = out[:, :, y0 - yp0:y0 - yp0 + (y1 - y0), x0 - xp0:x0 - xp0 + (x1 - x0)]
    check("tiled", got, ref[:, :, :75, :88].cpu(), 1e-5)


def test_training_trajectory_matches_oracle(dev):
    """Five AdamW steps of the tiny NAFNet on a fixed batch (L1 loss): the HIP path and the CPU oracle must follow the same
    loss trajectory -- forward, backward and the parameter gradients feeding the optimizer all agree step after step."""
    net = _build_net(TINY, dev)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = keyed_input("traj.x", (2, 3, 32, 32))
    gt = keyed_input("traj.gt", (2, 3, 32, 32))
    kw = dict(lr=2e-3, betas=(0.9, 0.9), weight_decay=0.0)
    opt_g = torch.optim.AdamW(net.parameters(), **kw)
    opt_r = torch.optim.AdamW([P[k] for k in sd], **kw)
    xg, gg = x.to(dev), gt.to(dev)
    lg, lr_ = [], []
    for _ in range(5):
        opt_g.zero_grad(set_to_none=True)
        l = (net(xg) - gg).abs().mean()
        l.backward()
        opt_g.step()
        lg.append(float(l))
        opt_r.zero_grad(set_to_none=True)
        yr, _ = O.nafnet_forward(x, P)
        l2 = (yr - gt).abs().mean()
        l2.backward()
        opt_r.step()
        lr_.append(float(l2))
    assert lr_[-1] < lr_[0], "the oracle itself must be learning"
    # the first steps agree to ~1e-7; from the fourth step on Adam's normalised update amplifies rounding-level gradient
    # differences (summation order) by ~30x per step on this tiny random net, which is why the test stops at five steps
    for i, (a, b) in enumerate(zip(lg, lr_)):
        assert abs(a - b) <= (2e-6 if i < 3 else 1e-4) * abs(b), (i, lg, lr_)
    for k in sd:
        check("param " + k, net.state_dict()[k], P[k].detach(), 5e-3)   # lr = 2e-3: an update whose sign flips shows up as ~4e-3


@pytest.mark.gpu
def test_fused_narrow_level_inference_drops_saved_tensors(dev):
    """Where the forward 1 x 1 chains are fused (dcpt_nafblock_fused_ffn(C) = 1: C = 64, ffn_f32.hip) LN1(inp) / LN2(y) / the gate are
    never allocated, and a forward that no backward follows passes no v / statistics buffers either; its output is bit-identical to
    the training forward's (ragged last group of 32 pixels included)."""
    from dcpt_amd import _lib, functional as DF

    lib = _lib.load()
    assert lib.dcpt_nafblock_fused_ffn(64) == 1 and lib.dcpt_nafblock_fused_ffn(128) == 0
    for shape in [(3, 64, 5, 7), (2, 64, 32, 32)]:
        P = block_params(64, "inf64.")
        Pd = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
        x = keyed_input("inf64.x", shape, lo=-1.0, hi=1.0).to(dev).contiguous(memory_format=torch.channels_last)
        y_train = DF.nafblock(x.clone().requires_grad_(True), {fk: Pd[rk] for fk, rk in FUSED.items()})
        assert DF._NAFBlockFn.last_infer is False
        with torch.no_grad():   # parameters that require grad (net_g in validation): the grad MODE selects the inference form
            y_inf = DF.nafblock(x, {fk: Pd[rk] for fk, rk in FUSED.items()})
        assert DF._NAFBlockFn.last_infer is True   # the NULL v / statistics form of dcpt_nafblock_fwd ran
        torch.cuda.synchronize()
        assert torch.equal(y_train.detach(), y_inf)
