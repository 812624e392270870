"""GPU parity: Restormer MDTA / GDFN blocks, the NHWC helpers and the tiny Restormer / Restormer_origin networks vs
the golden vectors of the real reference and vs the oracle on further shapes."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict, keyed_tensor
from oracle import restormer_oracle as R

pytestmark = pytest.mark.gpu
R_CFG = dict(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8])


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


def _block(dev, lnt, dim, heads):
    from basicsr.archs.restormer_arch import TransformerBlock

    return TransformerBlock(dim, heads, 2.66, False, lnt)


@pytest.mark.parametrize("lnt", ["BiasFree", "WithBias"])
def test_block_golden(dev, golden_dir, lnt):
    g = np.load(os.path.join(golden_dir, f"restormer_block_{lnt}.npz"))
    blk = _block(dev, lnt, 48, 1)
    sd = {k: keyed_tensor(f"tb{lnt}." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(dev)
    x = keyed_input(f"tb{lnt}.x", (2, 48, 12, 10), lo=-1.0, hi=1.0).to(dev).requires_grad_(True)
    go = keyed_input(f"tb{lnt}.go", (2, 48, 12, 10), lo=-1.0, hi=1.0).to(dev)
    y = blk(x)
    y.backward(go)
    check("y", y, g["y"], 5e-5)
    check("dx", x.grad, g["dx"], 2e-4)
    for k, p in blk.named_parameters():
        check("grad " + k, p.grad, g["g." + k], 3e-4)


@pytest.mark.parametrize("lnt,dim,heads,B,H,W", [("BiasFree", 32, 2, 1, 9, 7), ("WithBias", 16, 4, 3, 4, 6),
                                                  ("BiasFree", 96, 1, 1, 16, 16), ("BiasFree", 64, 8, 2, 8, 8)])
def test_block_oracle(dev, lnt, dim, heads, B, H, W):
    blk = _block(dev, lnt, dim, heads)
    sd = {k: keyed_tensor(f"ob{lnt}{dim}." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd, strict=True)
    x = keyed_input("ob.x", (B, dim, H, W), lo=-1.0, hi=1.0)
    go = keyed_input("ob.go", (B, dim, H, W), lo=-1.0, hi=1.0)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = R.transformer_block(xr, P, "")
    yr.backward(go)
    blk = blk.to(dev)
    xg = x.to(dev).requires_grad_(True)
    y = blk(xg)
    y.backward(go.to(dev))
    check("y", y, yr, 5e-5)
    check("dx", xg.grad, xr.grad, 2e-4)
    for k, p in blk.named_parameters():
        check("grad " + k, p.grad, P[k].grad, 3e-4)


def test_helpers(dev):
    from dcpt_amd import functional as DF

    x = keyed_input("h.x", (2, 8, 6, 10), lo=-1, hi=1)
    xg = x.to(dev).requires_grad_(True)
    y = DF.pixel_unshuffle2(xg)
    assert torch.equal(y.cpu(), F.pixel_unshuffle(x, 2))
    z = DF.pixel_shuffle2(y)
    assert torch.equal(z.cpu(), x)
    z.backward(torch.ones_like(z))
    assert torch.equal(xg.grad.cpu(), torch.ones_like(x))
    a, b = keyed_input("h.a", (2, 8, 4, 4)), keyed_input("h.b", (2, 12, 4, 4))
    ag, bg = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    c = DF.concat_channels(ag, bg)
    assert torch.equal(c.cpu(), torch.cat([a, b], 1))
    go = keyed_input("h.go", (2, 20, 4, 4))
    c.backward(go.to(dev))
    assert torch.equal(ag.grad.cpu(), go[:, :8]) and torch.equal(bg.grad.cpu(), go[:, 8:])
    for ks, (ci, co) in ((1, (16, 8)), (3, (8, 24))):
        xi = keyed_input("h.cx", (2, ci, 5, 7), lo=-1, hi=1)
        w = keyed_tensor("h.conv.weight", (co, ci, ks, ks))
        gy = keyed_input("h.cgo", (2, co, 5, 7), lo=-1, hi=1)
        xr, wr = xi.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, padding=ks // 2)
        yr.backward(gy)
        xq, wq = xi.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        yq = DF.conv_nobias(xq, wq)
        yq.backward(gy.to(dev))
        check("conv y", yq, yr, 1e-5)
        check("conv dx", xq.grad, xr.grad, 1e-5)
        check("conv dw", wq.grad, wr.grad, 1e-5)


@pytest.mark.parametrize("tag,name", [("restormer", "Restormer"), ("restormer_origin", "Restormer_origin")])
def test_restormer_tiny_golden(dev, golden_dir, tag, name):
    from basicsr.archs import build_network

    g = np.load(os.path.join(golden_dir, f"{tag}_tiny.npz"))
    net = build_network(dict(type=name, **R_CFG))
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net.load_state_dict(keyed_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0), strict=True)
    net = net.to(dev)
    x = keyed_input(f"{tag}.x", (2, 3, 32, 32)).to(dev).requires_grad_(True)
    gw = keyed_input(f"{tag}.gw", (2, 3, 32, 32), lo=-1.0, hi=1.0).to(dev)
    taps = []
    if name == "Restormer":
        hooks = [m.register_forward_hook(lambda mod, i, o: taps.append(o)) for n, m in net.named_modules()
                 if "decoder_level" in n and n.count(".") == 1]
        assert len(hooks) == 3
    y = net(x)
    (y * gw).sum().backward()
    check("y", y, g["y"], 2e-4)
    check("dx", x.grad, g["dx"], 1e-3)
    params = dict(net.named_parameters())
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(params[n].grad.double().pow(2).sum().sqrt())
        assert abs(mine - l2) <= 2e-3 * max(1e-7, l2), (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names":
            check("grad " + k[2:], params[k[2:]].grad, g[k], 2e-3)
    if name == "Restormer":
        assert [tuple(t.shape) for t in taps] == [(2, 64, 8, 8), (2, 32, 16, 16), (2, 32, 32, 32)]
        assert net(x.detach(), hook=True) is None


def test_directional_derivative_full_size(dev):
    """Size-independent property at BASELINE.json's Restormer configuration (defaults: dim 48, [4,6,6,8], B = 64, 128 x 128):
    analytic directional derivative of the whole backward pass == central difference of two forward passes (smooth loss, fp64
    reduction); see tests/test_gpu_parity.py::test_directional_derivative_full_size."""
    from basicsr.archs import build_network

    net = build_network(dict(type="Restormer"))
    net.load_state_dict(keyed_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0), strict=True)
    net = net.to(dev)
    x = torch.rand((64, 3, 128, 128), generator=torch.Generator().manual_seed(21)).to(dev).requires_grad_(True)
    gt = torch.rand((64, 3, 128, 128), generator=torch.Generator().manual_seed(22)).to(dev)

    def loss_of(inp):
        return (net(inp).double() - gt.double()).pow(2).mean()

    loss = loss_of(x)
    loss.backward()
    gen = torch.Generator().manual_seed(23)
    params = list(net.parameters())
    dirs = [torch.randn(p.shape, generator=gen).to(dev) * p.detach().abs().mean().clamp_min(1e-3) for p in params]
    dx = torch.randn(x.shape, generator=gen).to(dev) * 0.1
    analytic = float(sum((p.grad.double() * d.double()).sum() for p, d in zip(params, dirs)) + (x.grad.double() * dx.double()).sum())
    eps, vals = 1e-3, []
    with torch.no_grad():
        for sign in (+1.0, -1.0):
            for p, d in zip(params, dirs):
                p.add_(d, alpha=sign * eps)
            vals.append(float(loss_of(x.detach() + sign * eps * dx)))
            for p, d in zip(params, dirs):
                p.sub_(d, alpha=sign * eps)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(analytic) > 1e-4, analytic
    assert abs(numeric - analytic) <= 3e-2 * abs(analytic), (numeric, analytic, float(loss))


def test_saved_tensor_modes_equal_full(dev):
    """The save modes of the Restormer halves (functional.set_restormer_save / DCPT_RESTORMER_SAVE) -- "balanced" (LN(x), attn @ v and
    the GDFN gate product recomputed in backward; the DEFAULT: it does not depend on free memory), "lean" (also the qkv conv output) and
    the opt-in "auto" (one of the three per network forward, by the device memory in use when it starts) -- give bit-identical outputs
    and gradients to "full": the recomputation runs the same kernels on the same inputs; each keeps fewer bytes alive between
    forward and backward than the one before."""
    from basicsr.archs import build_network
    from dcpt_amd import functional as DF

    shapes = {k: tuple(v.shape) for k, v in build_network(dict(type="Restormer", **R_CFG)).state_dict().items()}
    sd = keyed_state_dict(shapes, seed=0)
    x = keyed_input("lean.x", (2, 3, 32, 32)).to(dev)
    res = {}
    prev = DF.set_restormer_save("full")
    assert prev == "balanced"   # the default: explicit, the same on every rank / next to any co-resident model
    try:
        for mode in ("full", "balanced", "lean", "auto"):
            DF.set_restormer_save(mode)
            net = build_network(dict(type="Restormer", **R_CFG))
            net.load_state_dict(sd, strict=True)
            net = net.to(dev)
            torch.cuda.synchronize()
            base = torch.cuda.memory_allocated()
            y = net(x)
            torch.cuda.synchronize()
            held = torch.cuda.memory_allocated() - base
            y.square().mean().backward()
            torch.cuda.synchronize()
            res[mode] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}, held)
            del net, y
    finally:
        DF.set_restormer_save(prev)
    for mode in ("balanced", "lean", "auto"):
        assert torch.equal(res["full"][0], res[mode][0])
        for k in res["full"][1]:
            assert torch.equal(res["full"][1][k], res[mode][1][k]), (mode, k)
    assert res["balanced"][2] < 0.85 * res["full"][2], (res["balanced"][2], res["full"][2])
    assert res["lean"][2] < 0.7 * res["full"][2], (res["lean"][2], res["full"][2])
    # the mode is also a per-network key (``network_g.save_mode``): it scopes one forward and wins over the process default
    net = build_network(dict(type="Restormer", save_mode="lean", **R_CFG))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    y = net(x)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - base
    assert torch.equal(y.detach(), res["full"][0]) and abs(held - res["lean"][2]) <= 0.02 * res["lean"][2], (held, res["lean"][2])
    with pytest.raises(ValueError):
        build_network(dict(type="Restormer", save_mode="everything", **R_CFG))
