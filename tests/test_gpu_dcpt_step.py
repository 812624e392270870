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
