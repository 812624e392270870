"""Dataset plumbing for ``test.py`` (reference basicsr/data/__init__.py:33-118).  The reference's IO stack
(cv2, lmdb, on-the-fly degradations) is out of scope; what the hot path needs from it
is a dict with ``lq``/``gt`` float32 CHW tensors in [0,1] (+ ``dataset_idx``).  A minimal PIL reader covers
paired folders; when a configured root does not exist (the YAMLs point at /mnt/nasv3/...), a seeded synthetic
pair stands in so the CLI still runs end to end."""
from copy import deepcopy
from os import path as osp

import numpy as np
import torch
from torch.utils import data as tdata

from basicsr.utils import get_root_logger, scandir
from basicsr.utils.registry import DATASET_REGISTRY

__all__ = ["build_dataset", "build_dataloader"]
_IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".PNG", ".JPG")


def _read_rgb(path):
    from PIL import Image

    with Image.open(path) as im:
        return torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()


@DATASET_REGISTRY.register()
class SyntheticPairedDataset(tdata.Dataset):
    """``num`` seeded image pairs of ``size`` (gt = smooth random field, lq = gt + noise)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.num = int(opt.get("num", 1))
        size = opt.get("size", 256)
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.seed = int(opt.get("seed", 0))
        self.sigma = float(opt.get("sigma_range", 25)) / 255.0

    def __len__(self):
        return self.num

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 100003 + index)
        h, w = self.size
        low = torch.rand((1, 3, max(2, h // 16), max(2, w // 16)), generator=g)
        gt = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0].clamp(0, 1)
        lq = (gt + self.sigma * torch.randn((3, h, w), generator=g)).clamp(0, 1)
        name = f"synthetic_{index:04d}"
        return {"lq": lq, "gt": gt, "lq_path": name, "gt_path": name}


@DATASET_REGISTRY.register()
class PairedImageDataset(tdata.Dataset):
    """Paired folders with identical file names (reference basicsr/data/paired_image_dataset.py:25-192,
    test phase only: no crops / augmentation)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.gt_root, self.lq_root = opt["dataroot_gt"], opt.get("dataroot_lq") or opt["dataroot_gt"]
        self.names = sorted(f for f in scandir(self.gt_root) if f.endswith(_IMG_EXT))
        self.sigma = opt.get("sigma_range") if opt.get("dataroot_lq") is None else None

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        name = self.names[index]
        gt = _read_rgb(osp.join(self.gt_root, name))
        if self.sigma is not None:  # PairedImageDenoiseDataset: seeded Gaussian noise (paired_image_dataset.py:397-402)
            rs = np.random.RandomState(0)
            lq = gt + torch.from_numpy(rs.normal(0, self.sigma / 255.0, tuple(gt.shape)).astype(np.float32))
        else:
            lq = _read_rgb(osp.join(self.lq_root, name))
        return {"lq": lq, "gt": gt, "lq_path": osp.join(self.lq_root, name), "gt_path": osp.join(self.gt_root, name)}


for _alias in ("PairedImageDenoiseDataset", "PairedImageDehazeDataset"):
    DATASET_REGISTRY._obj_map[_alias] = PairedImageDataset


def build_dataset(dataset_opt):
    dataset_opt = deepcopy(dataset_opt)
    logger = get_root_logger()
    root = dataset_opt.get("dataroot_gt")
    if dataset_opt["type"] != "SyntheticPairedDataset" and (root is None or not osp.isdir(root)):
        logger.warning(f"Dataset root {root} of [{dataset_opt.get('name')}] not found: using a seeded synthetic pair instead.")
        dataset = SyntheticPairedDataset(dataset_opt)
    else:
        dataset = DATASET_REGISTRY.get(dataset_opt["type"])(dataset_opt)
    logger.info(f"Dataset [{dataset.__class__.__name__}] - {dataset_opt['name']} is built.")
    return dataset


def build_dataloader(dataset, dataset_opt, num_gpu=1, dist=False, sampler=None, seed=None):
    phase = dataset_opt.get("phase", "test")
    if phase in ("val", "test"):
        return tdata.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=0)
    raise ValueError(f"Wrong dataset phase: {phase}. Supported ones are 'val' and 'test' (the reference ships no train loop).")
