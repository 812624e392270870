"""DCPT pre-training step and its variants (reference
basicsr/models/degradation_classification_pretrain_model.py:16-169 ``DCPTModel``,
..._direct_train_model.py ``DCTModel`` (reconstruction forward on ``lq``, :140) and
..._model.py ``DCModel`` (frozen encoder under no_grad, detached taps, :94-97,:122-148)).

One step = encoder forward on the clean image (pixel loss), encoder forward on the degraded image with
forward hooks on the decoder groups, classifier head on the tapped features (classification loss), ONE
backward through head -> taps -> encoder (twice), two optimizer steps.  Under DDP every fused block returns
all its gradients at once, so the bucketed RCCL all-reduce overlaps the remaining backward kernels; a
bucket is only complete after BOTH uses of each encoder weight have back-propagated.

By default (``train.batched_encoder_passes: true``) the two encoder forwards of a step run as ONE pass over the two batches stacked
(every encoder op is per-sample): same losses, logits and gradients up to fp32 summation order, half the launches, one
weight-gradient GEMM per layer; ``false`` runs the reference's two passes literally."""
from __future__ import annotations

from collections import OrderedDict

import torch

from basicsr.archs import build_network
from basicsr.losses import build_loss
from basicsr.utils import get_root_logger
from basicsr.utils.registry import MODEL_REGISTRY

from .base_model import BaseModel


def top1_accuracy(logits: torch.Tensor, target: torch.Tensor) -> float:
    """timm.utils.metrics.accuracy(..., topk=(1,)) restated: percentage of argmax hits."""
    return float((logits.argmax(dim=1) == target).float().mean().item() * 100.0)


@MODEL_REGISTRY.register()
class DCPTModel(BaseModel):
    recon_on_lq = False       # DCTModel: True
    freeze_encoder = False    # DCModel: True

    def __init__(self, opt):
        super().__init__(opt)
        self.net_g = self.model_to_device(build_network(opt["network_g"]))
        self.net_dc = self.model_to_device(build_network(opt["network_dc"]))
        for net, tag in ((self.net_g, "g"), (self.net_dc, "dc")):
            path = self.opt["path"].get(f"pretrain_network_{tag}", None)
            if path is not None:
                self.load_network(net, path, self.opt["path"].get(f"strict_load_{tag}", True),
                                  self.opt["path"].get(f"param_key_{tag}", "params"), self.opt.get("remove_norm", False))
        self.hook_outputs = []
        self.hooks = []
        self.batched_encoder_passes = bool((opt.get("train") or {}).get("batched_encoder_passes", True))
        if self.is_train:
            self.init_training_settings()

    # -- hooks -------------------------------------------------------------------------------
    def hook_forward_fn(self, module, input, output):  # noqa: A002
        head = ()
        if isinstance(output, tuple):
            head, output = output[:-1], output[-1]
        rng = getattr(self, "_tap_rows", None)
        if rng is not None and not self.freeze_encoder and torch.is_tensor(output) and output.is_cuda:
            # stacked encoder pass: the head taps samples rng[0]..rng[1]-1 of this feature map and the network goes on with all of it --
            # one autograd node for both uses, so that the two gradients are merged in place instead of zero-padded and added
            from dcpt_amd.functional import tap_split

            through, tap = tap_split(output, rng[0], rng[1])
            self.hook_outputs.append(tap)
            return head + (through,) if head else through   # (a tuple output stays a tuple downstream: only its last element is replaced)
        self.hook_outputs.append(output.detach() if self.freeze_encoder else output)
        return None

    def install_hooks(self):
        hook_names = self.opt.get("hook_names", None)
        if hook_names is None:
            raise ValueError("hook_names is required (e.g. 'decoder' for NAFNet, 'decoder_level' for Restormer)")
        # reference :65-68 (it walks the wrapped net, whose names gain a "module." prefix under DDP and then never match)
        for name, module in self.get_bare_model(self.net_g).named_modules():
            if hook_names in name and name.count(".") == 1:
                self.hooks.append(module.register_forward_hook(self.hook_forward_fn))
        if not self.hooks:
            raise ValueError(f"no module of net_g matches hook_names={hook_names!r}")

    def init_training_settings(self):
        self.net_g.train()
        self.net_dc.train()
        train_opt = self.opt["train"]
        self.install_hooks()
        self.cri_classify = build_loss(train_opt["classify_opt"]).to(self.device) if train_opt.get("classify_opt") else None
        self.cri_pixel = build_loss(train_opt["pixel_opt"]).to(self.device) if train_opt.get("pixel_opt") else None
        if self.cri_classify is None:
            raise ValueError("Classify loss is None.")
        self.setup_optimizers()
        if train_opt.get("scheduler"):
            self.setup_schedulers()

    def setup_optimizers(self):
        train_opt = self.opt["train"]
        self.optimizer_g = None
        nets = [("dc", self.net_dc)] if self.freeze_encoder else [("g", self.net_g), ("dc", self.net_dc)]
        for tag, net in nets:
            params = []
            for k, v in net.named_parameters():
                if v.requires_grad:
                    params.append(v)
                else:
                    get_root_logger().warning(f"Params {k} will not be optimized.")
            cfg = dict(train_opt[f"optim_{tag}"])
            opt = self.get_optimizer(cfg.pop("type"), params, **cfg)
            setattr(self, f"optimizer_{tag}", opt)
            self.optimizers.append(opt)

    def feed_data(self, data):
        self.lq = data["lq"].to(self.device, non_blocking=True)
        self.dataset_idx = data["dataset_idx"].to(self.device, non_blocking=True)
        if "gt" in data:
            self.gt = data["gt"].to(self.device, non_blocking=True)

    # -- the step ----------------------------------------------------------------------------------
    def optimize_parameters(self, current_iter):
        loss_dict = OrderedDict()
        l_total = 0
        taps = None
        if not self.freeze_encoder:
            self.net_g.train()
            self.net_dc.eval()
            self.optimizer_g.zero_grad()
            recon_in = self.lq if self.recon_on_lq else self.gt
            if self.batched_encoder_passes and recon_in.shape == self.lq.shape:
                # The reference runs the encoder twice per step (reconstruction input, then the degraded image for the taps).  Every
                # encoder op is per-sample, so ONE pass over the two batches stacked gives the same activations and -- the weight
                # gradients being sums over samples -- the same gradients up to fp32 summation order, with half the launches, one
                # weight-gradient GEMM per layer instead of two and no gradient-accumulation adds (train.batched_encoder_passes: false
                # restores the two passes).  DCTModel reconstructs from lq itself: one pass over the batch serves both.
                nb = self.lq.shape[0]
                self.hook_outputs = []
                if self.recon_on_lq:
                    pix_output = self.net_g(self.lq, hook=False)
                    taps = self.hook_outputs
                else:
                    from dcpt_amd.functional import take_batch   # (stride-preserving slice gradients: no NCHW round trips)

                    split = self.lq.is_cuda   # (the hooks hand back the second half of each tap themselves: hook_forward_fn)
                    self._tap_rows = (nb, 2 * nb) if split else None
                    try:
                        pix_output = take_batch(self.net_g(torch.cat([recon_in, self.lq], 0), hook=False), 0, nb)
                    finally:
                        self._tap_rows = None
                    # one stacked forward must have fired every hook exactly once, on the full 2B batch
                    want = nb if split else 2 * nb
                    if len(self.hook_outputs) != len(self.hooks) or any(t.shape[0] != want for t in self.hook_outputs):
                        raise RuntimeError(f"batched encoder pass: expected {len(self.hooks)} taps of batch {want}, got "
                                           f"{[tuple(t.shape) for t in self.hook_outputs]}")
                    taps = self.hook_outputs if split else [take_batch(t, nb, 2 * nb) for t in self.hook_outputs]
            else:
                pix_output = self.net_g(recon_in, hook=False)
            self.hook_outputs = []  # drop the taps recorded by the reconstruction forward
            if self.cri_pixel:
                l_pix = self.cri_pixel(pix_output, self.gt)
                l_total = l_total + l_pix
                loss_dict["l_pix"] = l_pix
        self.net_dc.train()
        self.optimizer_dc.zero_grad()
        if taps is not None:
            self.hook_outputs = taps
        elif self.freeze_encoder:
            self.net_g.eval()
            self.hook_outputs = []
            with torch.no_grad():
                self.net_g(self.lq, hook=True)
        else:
            self.net_g(self.lq, hook=True)  # returns None; the hooks collect decoder0..3
        cls_output = self.net_dc(self.lq, self.hook_outputs[::-1])
        l_classify = self.cri_classify(cls_output, self.dataset_idx)
        l_total = l_total + l_classify
        loss_dict["l_classify"] = l_classify
        l_total.backward()
        if self.optimizer_g is not None:
            self.optimizer_g.step()
        self.optimizer_dc.step()
        self.hook_outputs = []
        self.cls_output = cls_output.detach()
        self.log_dict = self.reduce_loss_dict(loss_dict)

    def test(self):
        if not self.hooks:
            self.install_hooks()
        self.net_g.eval()
        self.net_dc.eval()
        self.hook_outputs = []
        with torch.no_grad():
            self.net_g(self.lq, hook=True)
            self.cls_output = self.net_dc(self.lq, self.hook_outputs[::-1])
        self.hook_outputs = []

    def dist_validation(self, dataloader, current_iter, tb_logger, save_img, clamp=True):
        if self.opt["rank"] == 0:
            return self.nondist_validation(dataloader, current_iter, tb_logger, save_img, clamp)

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img, clamp=True):
        hits, n = 0.0, 0
        for val_data in dataloader:
            self.feed_data(val_data)
            self.test()
            hits += top1_accuracy(self.cls_output, self.dataset_idx) * self.lq.shape[0]
            n += self.lq.shape[0]
        acc = hits / max(1, n)
        self.metric_results = {"top1": acc}
        get_root_logger().info(f"Validation {dataloader.dataset.opt['name']}\n\t # top1: {acc:.4f}\n")
        return self.metric_results

    def save(self, epoch, current_iter):
        self.save_network(self.net_g, "net_g", current_iter)
        self.save_network(self.net_dc, "net_dc", current_iter)
        self.save_training_state(epoch, current_iter)


@MODEL_REGISTRY.register()
class DCTModel(DCPTModel):
    """direct-train variant: the reconstruction forward sees ``lq`` (reference ..._direct_train_model.py:140)."""
    recon_on_lq = True


@MODEL_REGISTRY.register()
class DCModel(DCPTModel):
    """classifier-only variant: frozen encoder under no_grad, detached taps (reference ..._model.py:94-97,:122-148)."""
    freeze_encoder = True
