"""GPU tests of the mechanisms around the kernels: the weight-gradient side stream must not change a single bit, the
LDS-DMA GEMM tiles must be exact for ragged shapes (rows / columns / reduction depth that do not fill a tile -- the
hardware range check supplies the zeros), and repeated calls must be bit-reproducible."""
import pytest
import torch
import torch.nn.functional as F

from dcpt_amd.keyed_init import keyed_input, keyed_tensor
from tests.test_gpu_parity import FUSED, block_params, check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dcpt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _block_grads(dev, c, shape, tag):
    from dcpt_amd import functional as DF

    P = block_params(c, tag)
    x = keyed_input(tag + "x", shape, lo=-1.0, hi=1.0)
    go = keyed_input(tag + "go", shape, lo=-1.0, hi=1.0)
    Pg = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
    xg = x.to(dev).requires_grad_(True)
    y = DF.nafblock(xg, {fk: Pg[rk] for fk, rk in FUSED.items()})
    y.backward(go.to(dev))
    torch.cuda.synchronize()
    return [y.detach().clone(), xg.grad.clone()] + [Pg[k].grad.clone() for k in sorted(Pg)]


@pytest.mark.parametrize("c,shape", [(32, (2, 32, 24, 20)), (128, (2, 128, 32, 32))])
def test_side_stream_is_bit_identical(dev, c, shape):
    from dcpt_amd import _lib

    lib = _lib.load()
    prev = lib.dcpt_set_side_stream(1)
    try:
        on = _block_grads(dev, c, shape, f"ss{c}.")
        on2 = _block_grads(dev, c, shape, f"ss{c}.")
        lib.dcpt_set_side_stream(0)
        off = _block_grads(dev, c, shape, f"ss{c}.")
    finally:
        lib.dcpt_set_side_stream(prev)
    for a, b, c_ in zip(on, on2, off):
        assert torch.equal(a, b), "two runs with the side stream differ (race?)"
        assert torch.equal(a, c_), "side stream on/off must give identical bits"


@pytest.mark.parametrize("B,H,W,Ci,Co", [(1, 16, 16, 16, 32), (2, 12, 20, 16, 16), (1, 9, 7, 36, 40), (1, 16, 16, 48, 200),
                                         (3, 5, 5, 132, 4), (1, 20, 13, 256, 260),
                                         # 96-wide GEMM tiles (chosen when they pad less than 128-wide ones and the grid is large)
                                         (3, 128, 128, 96, 288), (3, 128, 128, 192, 96), (3, 128, 128, 256, 96),
                                         (3, 128, 127, 100, 180), (3, 128, 128, 480, 96)])
def test_conv1x1_ragged_shapes(dev, B, H, W, Ci, Co):
    """forward (NT GEMM), input gradient (NT) and weight gradient (TN, split slabs) of a bias-free 1x1 conv"""
    from dcpt_amd import functional as DF

    x = keyed_input(f"rg{Ci}.{Co}.x", (B, Ci, H, W), lo=-1, hi=1)
    w = keyed_tensor(f"rg{Ci}.{Co}.conv.weight", (Co, Ci, 1, 1))
    go = keyed_input(f"rg{Ci}.{Co}.go", (B, Co, H, W), lo=-1, hi=1)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    yr.backward(go.double())
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = DF.conv_nobias(xg, wg)
    y.backward(go.to(dev))
    assert torch.isfinite(y).all()
    check("y", y, yr.float(), 2e-6)
    check("dx", xg.grad, xr.grad.float(), 2e-6)
    check("dw", wg.grad, wr.grad.float(), 5e-6)


def test_gemm_is_reproducible(dev):
    from dcpt_amd import functional as DF

    x = keyed_input("rep.x", (4, 96, 40, 24), lo=-1, hi=1).to(dev).requires_grad_(True)
    w = keyed_tensor("rep.conv.weight", (160, 96, 1, 1)).to(dev).requires_grad_(True)
    go = keyed_input("rep.go", (4, 160, 40, 24), lo=-1, hi=1).to(dev)
    outs = []
    for _ in range(3):
        x.grad = None
        w.grad = None
        y = DF.conv_nobias(x, w)
        y.backward(go)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_backward_on_a_user_stream_and_under_graph_capture(dev):
    """the side stream is keyed by the caller's stream, and is bypassed while that stream is being captured"""
    from dcpt_amd import functional as DF

    c, shape = 32, (2, 32, 16, 16)
    ref = _block_grads(dev, c, shape, "us.")
    st = torch.cuda.Stream(device=dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(st):
        got = _block_grads(dev, c, shape, "us.")
    torch.cuda.current_stream(dev).wait_stream(st)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)

    P = block_params(c, "us.")
    Pg = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
    x = keyed_input("us.x", shape, lo=-1.0, hi=1.0).to(dev).requires_grad_(True)
    go = keyed_input("us.go", shape, lo=-1.0, hi=1.0).to(dev)

    def run():
        y = DF.nafblock(x, {fk: Pg[rk] for fk, rk in FUSED.items()})
        gs = torch.autograd.grad(y, [x] + [Pg[k] for k in sorted(Pg)], go)
        return [y] + list(gs)

    st2 = torch.cuda.Stream(device=dev)
    st2.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(st2):
        for _ in range(2):
            run()      # warm-up: workspaces allocated before capture
    torch.cuda.current_stream(dev).wait_stream(st2)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = run()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(ref, outs):
        assert torch.equal(a, b.detach())


def test_large_image_windows(dev):
    """an image whose NHWC feature map exceeds 1 GiB per tensor (1536 x 1536 x 128 ch x 4 B = 1.2 GB): the depthwise and
    edge-conv kernels address it through per-row-range windows, the GEMMs through per-tile windows"""
    from dcpt_amd import functional as DF

    H = W = 1536
    c = 64
    x = (torch.rand((1, 3, H, W), device=dev) - 0.5)
    wi = keyed_tensor("big.intro.weight", (c, 3, 3, 3)).to(dev)
    bi = keyed_tensor("big.intro.bias", (c,)).to(dev)
    P = {k: v.to(dev) for k, v in block_params(c, "big.").items()}
    with torch.no_grad():
        f = DF.conv3x3_in(x, wi, bi)
        y = DF.nafblock(f, {fk: P[rk] for fk, rk in FUSED.items()})
        # reference on a crop whose receptive field (3x3 intro, 3x3 depthwise, global SCA pooling) we can reproduce: compare the
        # intro conv exactly and the block through a second run (determinism) + finiteness; the global pool forbids cropping
        r = F.conv2d(x[:, :, 700:900, 600:800], wi, bi, padding=1)[:, :, 1:-1, 1:-1]
        check("intro crop", f[:, :, 701:899, 601:799], r, 1e-5)
        y2 = DF.nafblock(f, {fk: P[rk] for fk, rk in FUSED.items()})
    assert torch.isfinite(y).all() and torch.equal(y, y2)
    assert float((y - f).abs().max()) > 0


def test_every_kernel_family_is_bit_stable_next_to_concurrent_bf16_gemms(dev):
    """tests/stream_stress.py as a test: every layer of the encoder (forward under no_grad; forward + backward of the blocks) in fp32 and
    bf16 storage runs 10 times on one stream while bf16 NAFBlocks run back to back on another, and must reproduce its quiet result bit for
    bit.  (Round 4: the ending conv did not -- a few elements per launch next to bf16 MFMA GEMMs, exact alone: packed-fp32 instructions with
    operand selection, LABNOTES.md 4h; the library is built without packed fp32 since.  Concurrent streams are what tiled inference, the weight-gradient side stream and DDP's all-reduce rely on.)"""
    from tests import stream_stress

    failed = stream_stress.scan(reps=10, verbose=False)
    assert not failed, f"not bit-stable next to concurrent GEMMs: {failed}"


@pytest.mark.parametrize("act", ["fp32", "bf16"])
def test_full_size_step_is_bit_identical_with_and_without_the_side_stream(dev, act):
    """BASELINE.json configs[1] at its real size: every parameter gradient of NAFNet-64, B = 32, 256 x 256 with the weight-gradient side
    stream equals the single-stream run bit for bit, twice (the small-block form of this test above cannot see rare per-element races)."""
    from basicsr.archs import build_network
    from dcpt_amd import _lib
    from dcpt_amd.keyed_init import fill_module_

    lib = _lib.load()
    full = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
    net = fill_module_(build_network(dict(type="NAFNetBaseline", act_dtype=act, **full))).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    lq, gt = torch.rand((32, 3, 256, 256), generator=g, device=dev), torch.rand((32, 3, 256, 256), generator=g, device=dev)

    def grads(side):
        lib.dcpt_set_side_stream(side)
        net.zero_grad(set_to_none=True)
        (net(lq) - gt).abs().mean().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net.parameters()]

    prev = lib.dcpt_set_side_stream(0)
    try:
        ref = grads(0)
        for _ in range(2):
            got = grads(1)
            bad = sum(not torch.equal(a, b) for a, b in zip(ref, got))
            assert bad == 0, f"{bad} of {len(ref)} parameter gradients differ between the two-stream and the single-stream step"
    finally:
        lib.dcpt_set_side_stream(prev)
