"""Caller-side base class of the hot path (reference basicsr/models/base_model.py): device placement and
the DDP wrap (:100-118), optimizer factory (:120-139), EMA (:86-95), checkpoint IO with the reference's
key conventions (:249-369), loss reduction to rank 0 (:432-457)."""
from __future__ import annotations

import os
import time
from collections import OrderedDict
from copy import deepcopy

import torch
from torch.nn.parallel import DataParallel, DistributedDataParallel

from basicsr.utils import get_root_logger
from dcpt_amd import functional as DF
from basicsr.utils.dist_util import master_only


class _PendingLog(OrderedDict):
    """losses of a step as device tensors, on their way into ``BaseModel.log_dict`` (reduce_loss_dict)"""


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt["num_gpu"] != 0 else "cpu")
        self.is_train = opt["is_train"]
        self.schedulers = []
        self.optimizers = []
        self.log_dict = OrderedDict()

    # -- protocol ---------------------------------------------------------------------------
    def feed_data(self, data):
        pass

    def optimize_parameters(self, current_iter):
        pass

    def get_current_visuals(self):
        pass

    def save(self, epoch, current_iter):
        pass

    def validation(self, dataloader, current_iter, tb_logger, save_img=False, clamp=True):
        if self.opt["dist"]:
            return self.dist_validation(dataloader, current_iter, tb_logger, save_img, clamp)
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img, clamp)

    def get_current_log(self):
        return self.log_dict

    # ``log_dict`` (reference base_model.py:448-457: an OrderedDict of Python floats, filled by every optimize_parameters) is materialised
    # LAZILY: a step stores its (reduced) loss tensors and the floats are made when somebody reads the log.  The reference's ``.item()`` per
    # loss per step is a host synchronisation at the end of every step -- the device then idles until the host has launched the next step's
    # first kernels (two gaps of 3.0 + 1.4 ms per all-bf16 DCPT step on MI355X, 4 % of it; profiles/r6/dcpt_allbf16_256_kernels_sync.txt);
    # a training loop reads the log every ``print_freq`` iterations, and only those iterations synchronise.  Same values, same type.
    @property
    def log_dict(self):
        pending = self.__dict__.get("_log_pending")
        if pending is not None:
            self.__dict__["_log_pending"] = None
            keys = list(pending.keys())
            vals = [pending[k] for k in keys]
            if vals and all(torch.is_tensor(v) for v in vals) and len({v.device for v in vals}) == 1:
                floats = torch.stack([v.detach().float().reshape(()) for v in vals]).tolist()   # one transfer, one synchronisation
            else:
                floats = [float(v) for v in vals]
            self.__dict__["_log_values"] = OrderedDict(zip(keys, floats))
        return self.__dict__.setdefault("_log_values", OrderedDict())

    @log_dict.setter
    def log_dict(self, value):
        if isinstance(value, _PendingLog):
            self.__dict__["_log_pending"] = value
        else:
            self.__dict__["_log_pending"] = None
            self.__dict__["_log_values"] = value

    # -- networks -----------------------------------------------------------------------------
    def model_to_device(self, net, dist=True, find_unused_parameters=None):
        """One process per GPU; gradients are averaged by DDP's bucketed all-reduce over RCCL/xGMI,
        overlapped with the remaining backward kernels (each fused block returns its gradients at once)."""
        net = net.to(self.device)
        if self.opt["dist"] and dist:
            if find_unused_parameters is None:
                find_unused_parameters = self.opt.get("find_unused_parameters", False)
            kwargs = dict(find_unused_parameters=find_unused_parameters)
            if self.device.type == "cuda":
                kwargs["device_ids"] = [torch.cuda.current_device()]
                kwargs["bucket_cap_mb"] = self.opt.get("ddp_bucket_cap_mb", 64)
                kwargs["gradient_as_bucket_view"] = True
            net = DistributedDataParallel(net, **kwargs)
            if self.device.type == "cuda":
                from dcpt_amd import ddp as dcpt_ddp

                dcpt_ddp.prepare(net)   # one divide per bucket, parameter gradients written straight into the bucket views
        elif self.opt["num_gpu"] > 1 and dist:
            net = DataParallel(net)
        return net

    def get_bare_model(self, net):
        return net.module if isinstance(net, (DataParallel, DistributedDataParallel)) else net

    def get_optimizer(self, optim_type, params, lr, **kwargs):
        table = {"Adam": torch.optim.Adam, "AdamW": torch.optim.AdamW, "Adamax": torch.optim.Adamax,
                 "SGD": torch.optim.SGD, "ASGD": torch.optim.ASGD, "RMSprop": torch.optim.RMSprop,
                 "Rprop": torch.optim.Rprop}
        if optim_type not in table:
            raise NotImplementedError(f"optimizer {optim_type} is not supported yet.")
        if optim_type == "AdamW" and kwargs.get("fused") and not kwargs.get("amsgrad"):
            # ``fused: true`` asks for a fused implementation: on fp32 CUDA parameters that is the library's multi-tensor kernel
            # (dcpt_amd/optim.py: torch's update rule and state-dict layout, a handful of launches for the 664 tensors of NAFNet-64)
            params = list(params)
            flat = [p for g in params for p in g["params"]] if params and isinstance(params[0], dict) else params
            if flat and all(p.is_cuda and p.dtype == torch.float32 for p in flat):
                from dcpt_amd.optim import FusedAdamW

                return FusedAdamW(params, lr, **kwargs)
        return table[optim_type](params, lr, **kwargs)

    def setup_schedulers(self):
        """reference :141-158: MultiStepLR / MultiStepRestartLR / CosineAnnealingRestartLR by name"""
        from . import lr_scheduler as lrs

        sched = dict(self.opt["train"]["scheduler"])
        kind = sched.pop("type")
        for optimizer in self.optimizers:
            if kind in ("MultiStepLR", "MultiStepRestartLR"):
                self.schedulers.append(lrs.MultiStepRestartLR(optimizer, **sched))
            elif kind == "CosineAnnealingRestartLR":
                self.schedulers.append(lrs.CosineAnnealingRestartLR(optimizer, **sched))
            else:
                raise NotImplementedError(f"Scheduler {kind} is not implemented yet.")

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        if current_iter > 1:
            for scheduler in self.schedulers:
                scheduler.step()
        if current_iter < warmup_iter:
            for optimizer in self.optimizers:
                for group in optimizer.param_groups:
                    group["lr"] = group["initial_lr"] / warmup_iter * current_iter

    def get_current_learning_rate(self):
        return [g["lr"] for g in self.optimizers[0].param_groups]

    # -- training state (reference :371-430) -------------------------------------------------------
    @master_only
    def save_training_state(self, epoch, current_iter):
        if current_iter == -1:
            return
        state = {"epoch": epoch, "iter": current_iter, "optimizers": [o.state_dict() for o in self.optimizers],
                 "schedulers": [s.state_dict() for s in self.schedulers]}
        path = os.path.join(self.opt["path"]["training_states"], f"{current_iter}.state")
        logger = get_root_logger()
        for attempt in range(3):
            try:
                torch.save(state, path)
                return
            except Exception as e:  # noqa: BLE001
                logger.warning(f"Save training state error: {e}, remaining retry times: {2 - attempt}")
                time.sleep(1)
        logger.warning(f"Still cannot save {path}. Just ignore it.")

    def resume_training(self, resume_state):
        opts, scheds = resume_state["optimizers"], resume_state["schedulers"]
        assert len(opts) == len(self.optimizers), "Wrong lengths of optimizers"
        assert len(scheds) == len(self.schedulers), "Wrong lengths of schedulers"
        for o, sd in zip(self.optimizers, opts):
            o.load_state_dict(sd)
        for s, sd in zip(self.schedulers, scheds):
            s.load_state_dict(sd)

    def model_ema(self, decay=0.999):
        src = dict(self.get_bare_model(self.net_g).named_parameters())
        dst = dict(self.net_g_ema.named_parameters())
        keys = list(dst.keys())
        with torch.no_grad():  # one multi-tensor launch pair instead of the reference's 664 mul_/add_ pairs
            torch._foreach_mul_([dst[k] for k in keys], decay)
            torch._foreach_add_([dst[k] for k in keys], [src[k].detach() for k in keys], alpha=1 - decay)
        # foreach / fused updates do not reliably bump the parameters' version counters: tell the kernels' weight caches
        # (dcpt_amd.functional.PackedWeightsBf16) explicitly that net_g_ema's parameters have new values
        DF.invalidate_packed_weights()

    # -- checkpoints ------------------------------------------------------------------------------
    @master_only
    def save_network(self, net, net_label, current_iter, param_key="params"):
        name = "latest" if current_iter == -1 else current_iter
        path = os.path.join(self.opt["path"]["models"], f"{net_label}_{name}.pth")
        nets = net if isinstance(net, list) else [net]
        keys = param_key if isinstance(param_key, list) else [param_key]
        assert len(nets) == len(keys), "The lengths of net and param_key should be the same."
        payload = {}
        for n, k in zip(nets, keys):
            sd = self.get_bare_model(n).state_dict()
            payload[k] = OrderedDict((kk[7:] if kk.startswith("module.") else kk, v.cpu()) for kk, v in sd.items())
        logger = get_root_logger()
        for attempt in range(3):
            try:
                torch.save(payload, path)
                return
            except Exception as e:  # noqa: BLE001
                logger.warning(f"Save model error: {e}, remaining retry times: {2 - attempt}")
                time.sleep(1)
        logger.warning(f"Still cannot save {path}. Just ignore it.")

    def _print_different_keys_loading(self, crt_net, load_net, strict=True):
        crt = self.get_bare_model(crt_net).state_dict()
        a, b = set(crt.keys()), set(load_net.keys())
        logger = get_root_logger()
        if a != b:
            logger.warning("Current net - loaded net:")
            for v in sorted(a - b):
                logger.warning(f"  {v}")
            logger.warning("Loaded net - current net:")
            for v in sorted(b - a):
                logger.warning(f"  {v}")
        if not strict:
            for k in a & b:
                if crt[k].size() != load_net[k].size():
                    logger.warning(f"Size different, ignore [{k}]: crt_net: {crt[k].shape}; load_net: {load_net[k].shape}")
                    load_net[k + ".ignore"] = load_net.pop(k)

    def load_network(self, net, load_path, strict=True, param_key="params", remove_norm=False):
        logger = get_root_logger()
        net = self.get_bare_model(net)
        load_net = torch.load(load_path, map_location="cpu")
        if param_key is not None:
            if param_key not in load_net and "params" in load_net:
                param_key = "params"
                logger.info("Loading: params_ema does not exist, use params.")
            load_net = load_net[param_key]
        logger.info(f"Loading {net.__class__.__name__} model from {load_path}, with param key: [{param_key}].")
        for k, v in deepcopy(load_net).items():
            if "norm" in k and remove_norm:
                continue
            if k.startswith("module."):
                load_net[k[7:]] = v
                load_net.pop(k)
        self._print_different_keys_loading(net, load_net, strict)
        net.load_state_dict(load_net, strict=strict)
        DF.invalidate_packed_weights()

    # -- logging ----------------------------------------------------------------------------------
    def reduce_loss_dict(self, loss_dict):
        with torch.no_grad():
            if self.opt["dist"]:
                keys = list(loss_dict.keys())
                losses = torch.stack([loss_dict[k] for k in keys], 0)
                torch.distributed.reduce(losses, dst=0)
                if self.opt["rank"] == 0:
                    losses /= self.opt["world_size"]
                loss_dict = {k: v for k, v in zip(keys, losses)}
            # (0-d device tensors, not floats: ``log_dict`` turns them into floats when the log is read -- no synchronisation in the step)
            return _PendingLog((k, v.detach().mean()) for k, v in loss_dict.items())
