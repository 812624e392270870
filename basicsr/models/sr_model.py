"""Restoration model: the caller of the encoder hot path (reference basicsr/models/sr_model.py):
train step (:132-174), test (:176-185), reflect padding to the window multiple (:244-271), tiled inference
(:273-361), validation with uint8 PSNR/SSIM (:375-499)."""
from __future__ import annotations

import math
from collections import OrderedDict
from os import path as osp

import numpy as np
import torch
from torch.nn import functional as F

from basicsr.archs import build_network
from basicsr.losses import build_loss
from basicsr.metrics import calculate_metric
from basicsr.utils import get_root_logger
from basicsr.utils.registry import MODEL_REGISTRY

from .base_model import BaseModel


def tensor2img_rgb(t: torch.Tensor) -> np.ndarray:
    """(1,C,H,W) or (C,H,W) float in [0,1] -> HWC uint8 (round), the pixel convention of tensor2img."""
    t = t.detach().float().cpu().clamp_(0, 1)
    if t.dim() == 4:
        t = t[0]
    return (t.permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8)


_TILE_STREAMS = {}   # device -> HIP streams of SRModel.test_tile (process-wide, see _tile_streams)


@MODEL_REGISTRY.register()
class SRModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.net_g = self.model_to_device(build_network(opt["network_g"]))
        self.grad_clip = opt.get("grad_clip", 0)
        self.scale = opt.get("scale", 1)
        self.mod_pad_h = self.mod_pad_w = 0
        load_path = self.opt["path"].get("pretrain_network_g", None)
        if load_path is not None:
            if osp.exists(load_path):
                self.load_network(self.net_g, load_path, self.opt["path"].get("strict_load_g", True),
                                  self.opt["path"].get("param_key_g", "params"), self.opt.get("remove_norm", False))
            elif self.opt.get("allow_missing_pretrain", False):
                get_root_logger().warning(f"pretrain_network_g {load_path} not found: keeping the initial weights.")
            else:
                raise FileNotFoundError(load_path)
        if self.is_train:
            self.init_training_settings()

    def init_training_settings(self):
        self.net_g.train()
        train_opt = self.opt["train"]
        self.ema_decay = train_opt.get("ema_decay", 0)
        if self.ema_decay > 0:
            self.net_g_ema = build_network(self.opt["network_g"]).to(self.device)
            self._init_ema()
            self.net_g_ema.eval()
        self.cri_pix = build_loss(train_opt["pixel_opt"]).to(self.device) if train_opt.get("pixel_opt") else None
        if self.cri_pix is None:
            raise ValueError("Both pixel and perceptual losses are None.")
        self.setup_optimizers()
        if train_opt.get("scheduler"):
            self.setup_schedulers()

    def _init_ema(self, only_if_checkpoint_has_ema=False):
        """reference sr_model.py:70-79: with a pretrained / resumed network the EMA copy is loaded from the checkpoint's
        ``params_ema`` (load_network falls back to ``params`` when the key is missing) so that a restart keeps the EMA
        history; only a fresh start copies net_g.  ``only_if_checkpoint_has_ema`` (DCDistModel, whose reference always starts the
        EMA as a copy of net_g -- which also keeps ``remove_norm`` / ``param_key_g`` consistent between the two): load only when
        the file really holds ``params_ema`` (a resumed run), copy net_g otherwise."""
        load_path = self.opt["path"].get("pretrain_network_g", None)
        if load_path is not None and osp.exists(load_path):
            if only_if_checkpoint_has_ema and "params_ema" not in torch.load(load_path, map_location="cpu"):
                self.model_ema(0)
                return
            self.load_network(self.net_g_ema, load_path, self.opt["path"].get("strict_load_g", True), "params_ema")
        else:
            self.model_ema(0)

    def setup_optimizers(self):
        train_opt = self.opt["train"]
        params = []
        for k, v in self.net_g.named_parameters():
            if v.requires_grad:
                params.append(v)
            else:
                get_root_logger().warning(f"Params {k} will not be optimized.")
        cfg = dict(train_opt["optim_g"])
        self.optimizer_g = self.get_optimizer(cfg.pop("type"), params, **cfg)
        self.optimizers.append(self.optimizer_g)

    def feed_data(self, data):
        self.lq = data["lq"].to(self.device, non_blocking=True)
        if "gt" in data:
            self.gt = data["gt"].to(self.device, non_blocking=True)

    def optimize_parameters(self, current_iter):
        self.net_g.train()
        self.optimizer_g.zero_grad()
        self.output = self.net_g(self.lq)
        loss_dict = OrderedDict()
        l_pix = self.cri_pix(self.output, self.gt)
        loss_dict["l_pix"] = l_pix
        l_pix.backward()
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.net_g.parameters(), self.grad_clip)
        self.optimizer_g.step()
        self.log_dict = self.reduce_loss_dict(loss_dict)
        if getattr(self, "ema_decay", 0) > 0:
            self.model_ema(decay=self.ema_decay)

    # -- inference -----------------------------------------------------------------------------
    def _net(self):
        return self.net_g_ema if hasattr(self, "net_g_ema") else self.net_g

    def test(self):
        net = self._net()
        was_training = net.training
        net.eval()
        with torch.no_grad():
            self.output = net(self.lq)
        if was_training and net is self.net_g:
            net.train()

    def test_selfensemble(self):
        """x8 geometric self-ensemble (reference sr_model.py:187-232; selected by ``ensemble: true`` at the top of the options,
        :398): the input under the 8 elements of the dihedral group (built as the reference builds its list: width flip "v", height
        flip "h", transpose "t", each applied to everything before it), the network on each, every output mapped back and the 8
        results averaged.  The flips / transposes stay on the device (the reference goes through numpy), and augmentations of
        equal shape run as one batch (the network is per-sample) up to a pixel budget: 1 pass for a small square image, 2 otherwise."""
        tf = {"v": lambda t: t.flip(-1), "h": lambda t: t.flip(-2), "t": lambda t: t.transpose(-1, -2)}
        lq_list = [self.lq]
        for op in ("v", "h", "t"):
            lq_list.extend([tf[op](t).contiguous() for t in lq_list])
        net = self._net()
        was_training = net.training
        net.eval()
        b = self.lq.shape[0]
        out_list = [None] * 8
        by_shape = {}
        for i, a in enumerate(lq_list):
            by_shape.setdefault(tuple(a.shape), []).append(i)
        # at most ``val.ensemble_max_pixels`` input pixels per pass (default 2^24 = what a 2K tiled pass already holds; the reference
        # runs one augmentation at a time, so a large validation image must not need 4-8x its activation memory here)
        budget = int((self.opt.get("val") or {}).get("ensemble_max_pixels", 1 << 24))
        with torch.no_grad():
            for idx in by_shape.values():
                per = max(1, budget // max(1, lq_list[idx[0]][:, 0].numel()))
                for c0 in range(0, len(idx), per):
                    part = idx[c0:c0 + per]
                    out = net(torch.cat([lq_list[i] for i in part], 0)) if len(part) > 1 else net(lq_list[part[0]])
                    for j, i in enumerate(part):
                        out_list[i] = out[j * b:(j + 1) * b]
        for i in range(8):   # undo in the reverse order of application (reference :218-226)
            if i > 3:
                out_list[i] = tf["t"](out_list[i])
            if i % 4 > 1:
                out_list[i] = tf["h"](out_list[i])
            if (i % 4) % 2 == 1:
                out_list[i] = tf["v"](out_list[i])
        self.output = torch.stack(out_list, 0).mean(dim=0)
        if was_training and net is self.net_g:
            net.train()

    def pre_test(self):
        """reflect-pad right/bottom to a multiple of ``network_g.window_size`` (sr_model.py:244-260)"""
        _, _, h, w = self.lq.size()
        self.mod_pad_h = self.mod_pad_w = 0
        if "window_size" not in self.opt["network_g"]:
            return
        ws = self.opt["network_g"]["window_size"]
        ws = max(ws) if isinstance(ws, (list, tuple)) else int(ws)
        self.scale = self.opt.get("scale", 1)
        if h % ws:
            self.mod_pad_h = ws - h % ws
        if w % ws:
            self.mod_pad_w = ws - w % ws
        self.lq = F.pad(self.lq, (0, self.mod_pad_w, 0, self.mod_pad_h), "reflect")

    def post_test(self):
        if "window_size" not in self.opt["network_g"]:
            return
        _, _, h, w = self.output.size()
        self.output = self.output[:, :, 0:h - self.mod_pad_h * self.scale, 0:w - self.mod_pad_w * self.scale]

    def test_tile(self):
        """independent ``infer_size`` tiles with ``tile_pad`` context, centre pasted back (sr_model.py:273-361).
        Same tiles, same arithmetic as the reference's loop; tiles of equal padded shape (interior / edge / corner
        classes) are stacked into one batch per class so the GPU sees a few large launches instead of many 1-image ones.

        ``tile.shard_across_ranks: true`` under a distributed launch (tiles are independent): batch number i goes to rank
        i % world_size, every rank pastes its tiles into a zero image and ONE reduce(SUM, dst=0) assembles the result on rank 0 -- the
        only exchange step of this path (the reference evaluates on rank 0 alone, sr_model.py:375-377)."""
        net = self._net()
        net.eval()
        size, pad, sc = self.opt["tile"]["infer_size"], self.opt["tile"]["tile_pad"], self.opt.get("scale", 1)
        max_batch = int(self.opt["tile"].get("max_batch", 16))
        b, c, height, width = self.lq.shape
        self.output = self.lq.new_zeros((b, c, height * sc, width * sc))
        groups = {}
        for ty in range(math.ceil(height / size)):
            for tx in range(math.ceil(width / size)):
                x0, y0 = tx * size, ty * size
                x1, y1 = min(x0 + size, width), min(y0 + size, height)
                xp0, yp0 = max(x0 - pad, 0), max(y0 - pad, 0)
                xp1, yp1 = min(x1 + pad, width), min(y1 + pad, height)
                groups.setdefault((yp1 - yp0, xp1 - xp0), []).append((x0, y0, x1, y1, xp0, yp0, xp1, yp1))
        shard = self._tile_shard()
        world, rank = (self.opt["world_size"], self.opt["rank"]) if shard else (1, 0)
        per = max(1, max_batch // b)
        batches = [tiles[i:i + per] for tiles in groups.values() for i in range(0, len(tiles), per)]
        batches = [chunk for k, chunk in enumerate(batches) if k % world == rank]

        def run(chunk):
            inp = torch.cat([self.lq[:, :, t[5]:t[7], t[4]:t[6]] for t in chunk], 0)
            with torch.no_grad():
                out = net(inp)
            for j, (x0, y0, x1, y1, xp0, yp0, xp1, yp1) in enumerate(chunk):
                ox, oy = (x0 - xp0) * sc, (y0 - yp0) * sc
                self.output[:, :, y0 * sc:y1 * sc, x0 * sc:x1 * sc] = \
                    out[j * b:(j + 1) * b, :, oy:oy + (y1 - y0) * sc, ox:ox + (x1 - x0) * sc]

        streams = self._tile_streams(len(batches), net)
        if not streams:
            for chunk in batches:
                run(chunk)
        else:
            # the shape classes are independent and each is too small to fill the chip (4 tiles of 544 x 544: 146 of 256 workgroups in the
            # level-3 GEMMs, hundreds of latency-bound small launches): one HIP stream per batch in flight, disjoint regions of the output
            cur = torch.cuda.current_stream(self.lq.device)
            for st in streams:
                st.wait_stream(cur)
            for k, chunk in enumerate(batches):
                with torch.cuda.stream(streams[k % len(streams)]):
                    run(chunk)
            for st in streams:
                cur.wait_stream(st)
        if shard:
            import torch.distributed as dist

            dist.reduce(self.output, dst=0, op=dist.ReduceOp.SUM)   # disjoint tiles: the sum is the assembled image (valid on rank 0)
        if net is self.net_g:
            net.train()

    def _tile_streams(self, nbatches, net):
        """HIP streams for the tile batches of ``test_tile`` (``tile.streams``, default 2 -- the measured optimum, 4 is slower again; 1 = the caller's stream only).  One stream when
        there is nothing to overlap, off the GPU, or when the network runs its GEMMs in the split-operand mode (one process-wide scratch
        buffer, dcpt_amd/csrc/gemm_x3.hip)."""
        n = min(int(self.opt["tile"].get("streams", 2)), nbatches)
        if n <= 1 or not self.lq.is_cuda:
            return []
        from dcpt_amd import functional as DF

        # (``net`` may be the DDP / DataParallel wrapper, which has no ``gemm_precision``: ask the bare network and every submodule)
        bare = self.get_bare_model(net)
        if DF.get_gemm_precision() != "fp32" or any(getattr(mod, "gemm_precision", None) not in (None, "fp32") for mod in bare.modules()):
            return []
        # one pool per device for the whole process: every torch.cuda.Stream() is another runtime stream, the runtime multiplexes them onto a
        # few hardware queues, and two streams that land on one queue do not overlap (a second model's own pair of streams measured 95 ms
        # where the first model's pair gave 83 on the same image)
        pool = _TILE_STREAMS.setdefault(self.lq.device, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(self.lq.device))
        return pool[:n]

    def _tile_shard(self):
        return bool(self.opt.get("dist")) and self.opt.get("world_size", 1) > 1 and "tile" in self.opt and \
            bool(self.opt["tile"].get("shard_across_ranks", False))

    def get_current_visuals(self):
        out = OrderedDict(lq=self.lq.detach().cpu(), result=self.output.detach().cpu())
        if hasattr(self, "gt"):
            out["gt"] = self.gt.detach().cpu()
        return out

    def dist_validation(self, dataloader, current_iter, tb_logger, save_img, clamp=True):
        if self.opt["rank"] == 0 or self._tile_shard():   # sharded tiles: every rank computes its tiles, rank 0 scores
            return self.nondist_validation(dataloader, current_iter, tb_logger, save_img, clamp)

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img, clamp=True):
        name = dataloader.dataset.opt["name"]
        metrics_opt = self.opt["val"].get("metrics") or {}
        results = OrderedDict((m, 0.0) for m in metrics_opt)
        n = 0
        for val_data in dataloader:
            self.feed_data(val_data)
            if self._tile_shard():
                # every rank built its own loader: a stochastic val set (sigma_type random / choice draws from Python's global
                # ``random``) could hand the ranks differently-noised inputs -- rank 0's sample is the one that is scored
                import torch.distributed as dist

                dist.broadcast(self.lq, src=0)
                if hasattr(self, "gt"):
                    dist.broadcast(self.gt, src=0)
            self.pre_test()
            if "tile" in self.opt:
                self.test_tile()
            elif self.opt.get("ensemble"):   # reference sr_model.py:396-399 (tile takes precedence)
                self.test_selfensemble()
            else:
                self.test()
            self.post_test()
            if self.opt.get("rank", 0) != 0:   # (tile-sharded validation: only rank 0 holds the assembled image)
                del self.lq, self.output
                if hasattr(self, "gt"):
                    del self.gt
                continue
            vis = self.get_current_visuals()
            # reference sr_model.py:408-413: clamp if asked; without clamp the metrics see the raw output (:421-430) and NaNs are zeroed
            # only afterwards (:431-436), i.e. for the saved image
            result = vis["result"].clamp(0, 1) if clamp else vis["result"]
            if save_img:
                result_img = result if clamp else torch.nan_to_num(result, nan=0.0)
                import os

                from PIL import Image

                stem = osp.splitext(osp.basename(val_data["lq_path"][0]))[0]
                folder = osp.join(self.opt["path"]["visualization"], name)
                os.makedirs(folder, exist_ok=True)
                Image.fromarray(tensor2img_rgb(result_img)).save(osp.join(folder, f"{stem}_{self.opt['name']}.png"))
            if "gt" in vis and metrics_opt:
                # the reference hands the float BCHW arrays in [0,1] to the metric, which quantises them to uint8 (:421-430)
                gt = (vis["gt"].clamp(0, 1) if clamp else vis["gt"]).numpy()
                for m, mo in metrics_opt.items():
                    results[m] += calculate_metric(dict(img=result.numpy(), img2=gt), mo)
            n += 1
            del self.lq, self.output
            if hasattr(self, "gt"):
                del self.gt
        for m in results:
            results[m] /= max(1, n)
        self.metric_results = results
        if self.opt.get("rank", 0) != 0:
            return results
        log = f"Validation {name}\n" + "".join(f"\t # {m}: {v:.4f}\n" for m, v in results.items())
        get_root_logger().info(log)
        if tb_logger:
            for m, v in results.items():
                tb_logger.add_scalar(f"metrics/{name}/{m}", v, current_iter)
        return results

    def save(self, epoch, current_iter):
        if hasattr(self, "net_g_ema"):
            self.save_network([self.net_g, self.net_g_ema], "net_g", current_iter, param_key=["params", "params_ema"])
        else:
            self.save_network(self.net_g, "net_g", current_iter)
        self.save_training_state(epoch, current_iter)
