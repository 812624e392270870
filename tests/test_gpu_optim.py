"""dcpt_amd.optim.FusedAdamW (include/dcpt_hip.h dcpt_adamw_step) against torch.optim.AdamW: the optimizer step of every training step of the
path (reference basicsr/models/base_model.py:70-93 builds torch.optim.AdamW from the YAML)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(7,), (64,), (1, 64, 1, 1), (128, 64, 1, 1), (1024, 512, 1, 1), (64, 1, 3, 3), (4099,), (3, 5, 7), (512, 512), (2, 4097)]


def _params(dev, seed=0, channels_last=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    ps = []
    for s in SHAPES:
        t = torch.randn(s, generator=g).to(dev)
        if channels_last and t.dim() == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t))
    return ps


def _grads(ps, seed):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    for p in ps:
        gr = (torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-3, 2, (1,), generator=g)))).to(p.device)
        p.grad = gr.contiguous(memory_format=torch.channels_last) if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) else gr


@pytest.mark.parametrize("kw", [dict(lr=1e-3, betas=(0.9, 0.9), weight_decay=0.0), dict(lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8),
                                dict(lr=1e-2, betas=(0.3, 0.6), weight_decay=0.1), dict(lr=1e-3, maximize=True)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_fused_adamw_matches_torch(kw, channels_last):
    from dcpt_amd.optim import FusedAdamW

    dev = torch.device("cuda", 0)
    a, b = _params(dev, 0, channels_last), _params(dev, 0, channels_last)
    oa, ob = FusedAdamW(a, **kw), torch.optim.AdamW(b, **kw)   # (torch's single-tensor path: the reference's optimizer)
    for step in range(6):
        _grads(a, step)
        _grads(b, step)
        oa.step()
        ob.step()
        for x, y in zip(a, b):
            # same formula, same operation order; torch divides by bias_correction2_sqrt where the kernel multiplies by its reciprocal
            torch.testing.assert_close(x, y, rtol=4e-6, atol=2e-7)
    for x, y in zip(a, b):
        # (an ulp here and there: the kernel's lerp / second-moment update contract into FMAs, six steps deep)
        for key in ("exp_avg", "exp_avg_sq"):   # (the first moment is a signed sum: elements near zero carry the rounding of the large terms)
            ma, mb = oa.state[x][key], ob.state[y][key]
            torch.testing.assert_close(ma, mb, rtol=2e-6, atol=2e-7 * float(mb.abs().max()))


def test_fused_adamw_state_dict_interchanges_with_torch():
    """a training state written by either optimizer resumes in the other (the CLI's save_training_state / resume_training round trip)"""
    from dcpt_amd.optim import FusedAdamW

    dev = torch.device("cuda", 0)
    kw = dict(lr=1e-3, betas=(0.9, 0.99), weight_decay=1e-2)
    a, b = _params(dev, 1), _params(dev, 1)
    oa, ob = FusedAdamW(a, **kw), torch.optim.AdamW(b, **kw)
    for step in range(3):
        _grads(a, step); _grads(b, step)
        oa.step(); ob.step()
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    assert sa["param_groups"][0].keys() >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"}
    assert all(isinstance(v["step"], torch.Tensor) and float(v["step"]) == 3.0 for v in sa["state"].values())
    # cross-load: ours <- torch's, torch <- ours; then three more steps must agree again
    a2, b2 = [torch.nn.Parameter(p.detach().clone()) for p in b], [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa2, ob2 = FusedAdamW(a2, **kw), torch.optim.AdamW(b2, **kw)
    oa2.load_state_dict(sb)
    ob2.load_state_dict(sa)
    for step in range(3, 6):
        _grads(a2, step); _grads(b2, step)
        oa2.step(); ob2.step()
    for x, y in zip(a2, b2):
        torch.testing.assert_close(x, y, rtol=4e-6, atol=2e-7)
    assert all(float(v["step"]) == 6.0 for v in oa2.state_dict()["state"].values())


def test_fused_adamw_skips_missing_grads_misaligned_views_and_storage_swaps():
    from dcpt_amd.optim import FusedAdamW

    dev = torch.device("cuda", 0)
    base_a, base_b = torch.randn(1 + 4099 + 64, device=dev), None
    base_b = base_a.clone()
    # parameters that are 4-byte-offset views of one buffer (DDP-style flat storage): the scalar path of the kernel
    a = [torch.nn.Parameter(base_a[1:1 + 4099]), torch.nn.Parameter(base_a[1 + 4099:]), torch.nn.Parameter(torch.ones(33, device=dev))]
    b = [torch.nn.Parameter(base_b[1:1 + 4099]), torch.nn.Parameter(base_b[1 + 4099:]), torch.nn.Parameter(torch.ones(33, device=dev))]
    oa, ob = FusedAdamW(a, lr=1e-2), torch.optim.AdamW(b, lr=1e-2)
    for step in range(4):
        for ps in (a, b):
            g = torch.Generator(device="cpu").manual_seed(step)
            for i, p in enumerate(ps):
                p.grad = None if (i == 2 and step % 2 == 0) else torch.randn(p.shape, generator=g).to(dev)   # the third one every other step only
        oa.step(); ob.step()
        if step == 1:   # the storage behind a parameter object is replaced (module.to(), p.data = ...): the next step must follow it
            a[1].data = a[1].data.clone()
            b[1].data = b[1].data.clone()
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=4e-6, atol=2e-7)
    assert float(oa.state_dict()["state"][2]["step"]) == 2.0 and float(oa.state_dict()["state"][0]["step"]) == 4.0


def test_base_model_builds_the_library_optimizer_for_fused_adamw():
    """``optim_g: {type: AdamW, fused: true}`` on CUDA parameters -> dcpt_amd.optim.FusedAdamW; without ``fused`` torch's own"""
    from basicsr.models.base_model import BaseModel
    from dcpt_amd.optim import FusedAdamW

    m = BaseModel.__new__(BaseModel)
    ps = [torch.nn.Parameter(torch.zeros(8, device="cuda"))]
    assert isinstance(m.get_optimizer("AdamW", ps, 1e-3, fused=True), FusedAdamW)
    assert type(m.get_optimizer("AdamW", ps, 1e-3)) is torch.optim.AdamW
    assert type(m.get_optimizer("AdamW", [{"params": ps}], 1e-3, fused=True, weight_decay=0.0)) is FusedAdamW
