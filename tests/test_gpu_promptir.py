"""GPU parity: PromptIR (reference basicsr/archs/promptir_arch.py) -- its transformer blocks (softmax attention, LayerNorm eps
1e-5), its prompt generation block (softmax-weighted prompt mix + bilinear resize + 3x3 conv) and the network, vs the golden
vectors of the real reference and vs the oracle on further shapes."""
import os

import numpy as np
import pytest
import torch

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict, keyed_tensor
from oracle import promptir_oracle as PO

pytestmark = pytest.mark.gpu
P_CFG = dict(num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
PG_SHAPES = {"prompt_param": (1, 5, 8, 6, 6), "linear_layer.weight": (5, 12), "linear_layer.bias": (5,), "conv3x3.weight": (8, 8, 3, 3)}


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


@pytest.mark.parametrize("lnt", ["BiasFree", "WithBias"])
def test_block_golden(dev, golden_dir, lnt):
    from basicsr.archs.promptir_arch import TransformerBlock

    g = np.load(os.path.join(golden_dir, f"promptir_block_{lnt}.npz"))
    blk = TransformerBlock(48, 2, 2.66, False, lnt)
    blk.load_state_dict({k: keyed_tensor(f"ptb{lnt}." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}, strict=True)
    blk = blk.to(dev)
    x = keyed_input(f"ptb{lnt}.x", (2, 48, 12, 10), lo=-1.0, hi=1.0).to(dev).requires_grad_(True)
    y = blk(x)
    y.backward(keyed_input(f"ptb{lnt}.go", (2, 48, 12, 10), lo=-1.0, hi=1.0).to(dev))
    check("y", y, g["y"], 5e-5)
    check("dx", x.grad, g["dx"], 2e-4)
    for k, p in blk.named_parameters():
        check("grad " + k, p.grad, g["g." + k], 3e-4)


@pytest.mark.parametrize("lnt,dim,heads,B,H,W", [("WithBias", 160, 4, 1, 9, 7), ("BiasFree", 320, 4, 2, 4, 6), ("WithBias", 704, 4, 1, 5, 3),
                                                  ("WithBias", 256, 1, 1, 8, 8)])
def test_block_oracle(dev, lnt, dim, heads, B, H, W):
    """PromptIR's noise-level block widths (160 / 320 / 704 channels, 4 heads -> 40 / 80 / 176 per head) and the 256-per-head
    limit of the softmax kernels"""
    from basicsr.archs.promptir_arch import TransformerBlock

    blk = TransformerBlock(dim, heads, 2.66, False, lnt)
    sd = {k: keyed_tensor(f"pbo{dim}." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd, strict=True)
    x = keyed_input(f"pbo{dim}.x", (B, dim, H, W), lo=-1.0, hi=1.0)
    go = keyed_input(f"pbo{dim}.go", (B, dim, H, W), lo=-1.0, hi=1.0)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = PO.transformer_block(xr, P, "")
    yr.backward(go)
    blk = blk.to(dev)
    xg = x.to(dev).requires_grad_(True)
    y = blk(xg)
    y.backward(go.to(dev))
    check("y", y, yr, 5e-5)
    check("dx", xg.grad, xr.grad, 3e-4)
    for k, p in blk.named_parameters():
        check("grad " + k, p.grad, P[k].grad, 5e-4)


def test_promptgen_golden(dev, golden_dir):
    from basicsr.archs.promptir_arch import PromptGenBlock

    g = np.load(os.path.join(golden_dir, "promptir_promptgen.npz"))
    pg = PromptGenBlock(prompt_dim=8, prompt_len=5, prompt_size=6, lin_dim=12)
    pg.load_state_dict({k: keyed_tensor("pg." + k, s) for k, s in PG_SHAPES.items()}, strict=True)
    pg = pg.to(dev)
    for tag, hw in (("up", (13, 9)), ("down", (4, 5)), ("same", (6, 6))):
        x = keyed_input(f"pg.{tag}.x", (3, 12) + hw, lo=-1.0, hi=1.0).to(dev).requires_grad_(True)
        pg.zero_grad()
        y = pg(x)
        y.backward(keyed_input(f"pg.{tag}.go", (3, 8) + hw, lo=-1.0, hi=1.0).to(dev))
        check(f"{tag} y", y, g[f"{tag}.y"], 1e-5)
        check(f"{tag} dx", x.grad, g[f"{tag}.dx"], 1e-4)
        for k, p in pg.named_parameters():
            check(f"{tag} grad {k}", p.grad, g[f"{tag}.g.{k}"], 1e-4)


def test_promptir_tiny_golden(dev, golden_dir):
    from basicsr.archs import build_network

    g = np.load(os.path.join(golden_dir, "promptir_tiny.npz"))
    net = build_network(dict(type="PromptIR", **P_CFG))
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net.load_state_dict(keyed_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0), strict=True)
    net = net.to(dev)
    x = keyed_input("pir.x", (2, 3, 64, 64)).to(dev).requires_grad_(True)
    y = net(x)
    (y * keyed_input("pir.gw", (2, 3, 64, 64), lo=-1.0, hi=1.0).to(dev)).sum().backward()
    check("y", y, g["y"], 2e-4)
    check("dx", x.grad, g["dx"], 1e-3)
    params = dict(net.named_parameters())
    for n, l2 in zip([str(s) for s in g["g_names"]], g["g_l2"]):
        mine = float(params[n].grad.double().pow(2).sum().sqrt())
        slack = 1e-3 if n.endswith(".temperature") else 0.0   # one scalar with heavy cancellation (see test_gpu_dcpt_step)
        assert abs(mine - l2) <= 2e-3 * max(1e-7, l2) + slack, (n, mine, l2)
    for k in g.files:
        if k.startswith("g.") and k != "g_names" and not k.endswith(".sub"):
            check("grad " + k[2:], params[k[2:]].grad, g[k], 2e-3)
    check("grad prompt1.prompt_param (subsampled)", params["prompt1.prompt_param"].grad[0, :, ::8, ::4, ::4], g["g.prompt1.prompt_param.sub"], 2e-3)
    assert net(x.detach(), hook=True) is None
    with torch.no_grad():
        check("y 40x24 (prompts resized down)", net(keyed_input("pir.xs", (1, 3, 40, 24)).to(dev)), g["y_small"], 2e-4)
        check("y 160x136 (prompts resized up)", net(keyed_input("pir.xl", (1, 3, 160, 136)).to(dev))[..., ::4, ::4], g["y_large"], 2e-4)
