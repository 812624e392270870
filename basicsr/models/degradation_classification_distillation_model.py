"""DC-guided fine-tuning of the restoration network (reference
basicsr/models/degradation_classification_distillation_model.py:21-185 ``DCDistModel``).

One step = ONE forward of the trainable ``net_g`` on the degraded image with forward hooks on its decoder taps, the FROZEN
(eval, requires_grad=False) classifier head ``net_dc`` on those taps, pixel loss + classification loss, one backward
(through the frozen head into the encoder), optional gradient clipping, one optimizer step, optional EMA (:152-185).
Inference and validation are SRModel's (pad to the window multiple, restore, crop; PSNR/SSIM on uint8)."""
from __future__ import annotations

from collections import OrderedDict
from os import path as osp

import torch

from basicsr.archs import build_network
from basicsr.losses import build_loss
from basicsr.utils import get_root_logger
from basicsr.utils.registry import MODEL_REGISTRY

from .sr_model import SRModel


def tap_modules(net, hook_names):
    """The modules the reference hooks (:81-88): children ``<group>.<i>`` of the groups whose name contains ``hook_names``,
    restricted there by hard-coded indices -- block 5 of ``*_level2`` / ``*_level3`` and block 3 of ``*_level1``, i.e. the LAST
    block of each group of the default ``Restormer_origin`` ([4, 6, 6, 8]).  Here: the last child of every matching group,
    which is the same set for that network and also defined for other depths and for NAFNet's ``decoder{i}``."""
    groups = OrderedDict()
    for name, module in net.named_modules():
        if hook_names in name and name.count(".") == 1 and name.split(".")[-1].isdigit():
            groups.setdefault(name.split(".")[0], []).append((int(name.split(".")[-1]), name, module))
    return [max(children)[1:] for children in groups.values()]


@MODEL_REGISTRY.register()
class DCDistModel(SRModel):
    def __init__(self, opt):
        self.net_dc = None
        super().__init__(opt)   # builds / loads net_g; calls init_training_settings() when training
        if self.net_dc is None:
            self._build_dc()

    def _build_dc(self):
        self.net_dc = self.model_to_device(build_network(self.opt["network_dc"]), dist=False)   # frozen: no gradients to reduce
        path = self.opt["path"].get("pretrain_network_dc", None)
        if path is not None:
            if osp.exists(path):
                self.load_network(self.net_dc, path, self.opt["path"].get("strict_load_dc", True),
                                  self.opt["path"].get("param_key_dc", "params"), self.opt.get("remove_norm", False))
            elif self.opt.get("allow_missing_pretrain", False):
                get_root_logger().warning(f"pretrain_network_dc {path} not found: keeping the initial weights.")
            else:
                raise FileNotFoundError(path)

    def init_training_settings(self):
        self._build_dc()
        self.net_dc.eval()
        for p in self.net_dc.parameters():
            p.requires_grad = False
        self.net_g.train()
        train_opt = self.opt["train"]
        self.grad_clip = self.opt.get("grad_clip", 0)
        self.ema_decay = train_opt.get("ema_decay", 0)
        if self.ema_decay > 0:
            get_root_logger().info(f"Use Exponential Moving Average with decay: {self.ema_decay}")
            self.net_g_ema = build_network(self.opt["network_g"]).to(self.device)
            self._init_ema(only_if_checkpoint_has_ema=True)   # the reference's DCDist model always starts from model_ema(0)
            self.net_g_ema.eval()
        self.hook_outputs, self.hooks = [], []
        hook_names = self.opt.get("hook_names", None)
        if hook_names is None:
            raise ValueError("hook_names is required (e.g. 'decoder_level' for Restormer_origin, 'decoder' for NAFNet)")
        for _, module in tap_modules(self.get_bare_model(self.net_g), hook_names):
            self.hooks.append(module.register_forward_hook(self.hook_forward_fn))
        if not self.hooks:
            raise ValueError(f"no module of net_g matches hook_names={hook_names!r}")
        self.cri_pixel = build_loss(train_opt["pixel_opt"]).to(self.device) if train_opt.get("pixel_opt") else None
        self.cri_classify = build_loss(train_opt["classify_opt"]).to(self.device) if train_opt.get("classify_opt") else None
        if self.cri_classify is None and self.cri_pixel is None:
            raise ValueError("Classify loss and Pixel loss are both None.")
        self.setup_optimizers()
        if train_opt.get("scheduler"):
            self.setup_schedulers()

    def hook_forward_fn(self, module, input, output):  # noqa: A002
        if not torch.is_grad_enabled():
            return   # inference (test / test_tile / validation run under no_grad): the taps are not used, do not keep them alive
        if isinstance(output, tuple):
            output = output[-1]
        self.hook_outputs.append(output)

    def feed_data(self, data):
        self.lq = data["lq"].to(self.device, non_blocking=True)
        if "dataset_idx" in data:
            self.dataset_idx = data["dataset_idx"].to(self.device, non_blocking=True)
        if "dataset_idx" in self.opt:   # one degradation for the whole run (:147-150)
            self.dataset_idx = torch.full((self.lq.shape[0],), int(self.opt["dataset_idx"]), dtype=torch.long, device=self.device)
        if "gt" in data:
            self.gt = data["gt"].to(self.device, non_blocking=True)

    def optimize_parameters(self, current_iter):
        self.net_dc.eval()
        self.net_g.train()
        self.optimizer_g.zero_grad()
        self.hook_outputs = []
        self.pix_output = self.net_g(self.lq)
        self.cls_output = self.net_dc(self.lq, self.hook_outputs[::-1])
        l_total = 0
        loss_dict = OrderedDict()
        if self.cri_pixel:
            l_pixel = self.cri_pixel(self.pix_output, self.gt)
            l_total = l_total + l_pixel
            loss_dict["l_pixel"] = l_pixel
        if self.cri_classify:
            l_classify = self.cri_classify(self.cls_output, self.dataset_idx)
            l_total = l_total + l_classify
            loss_dict["l_classify"] = l_classify
        l_total.backward()
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.net_g.parameters(), self.grad_clip)
        self.optimizer_g.step()
        self.hook_outputs = []
        self.log_dict = self.reduce_loss_dict(loss_dict)
        if self.ema_decay > 0:
            self.model_ema(decay=self.ema_decay)

    def test(self):
        """restoration only (:238-251; the reference leaves the classifier call commented out)"""
        super().test()
        self.pix_output = self.output
        self.hook_outputs = []
