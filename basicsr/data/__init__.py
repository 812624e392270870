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

__all__ = ["build_dataset", "build_dataloader", "paired_random_crop_augment"]
_IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".PNG", ".JPG")


def _read_rgb(path):
    from PIL import Image

    with Image.open(path) as im:
        return torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()


def _labelled(sample, opt):
    """a fixed degradation label for a stand-alone (validation) set: ``dataset_idx: k`` in its options"""
    if "dataset_idx" in opt:
        sample["dataset_idx"] = int(opt["dataset_idx"])
    return sample


def paired_random_crop_augment(lq, gt, opt):
    """train phase (reference basicsr/data/paired_image_dataset.py:152-166 with transforms.py paired_random_crop / augment,
    scale 1): the same random ``gt_size`` crop, horizontal flip and 90-degree rotation (vertical flip + transpose) of both
    images, drawn from Python's ``random`` like the reference."""
    import random

    size = opt.get("gt_size")
    if size:
        _, h, w = gt.shape
        if h < size or w < size:
            raise ValueError(f"image ({h}, {w}) is smaller than the training patch ({size}, {size})")
        top, left = random.randint(0, h - size), random.randint(0, w - size)
        lq, gt = lq[:, top:top + size, left:left + size], gt[:, top:top + size, left:left + size]
    hflip = opt.get("use_hflip", opt.get("use_flip", False)) and random.random() < 0.5
    vflip = opt.get("use_rot", False) and random.random() < 0.5
    rot90 = opt.get("use_rot", False) and random.random() < 0.5
    out = []
    for t in (lq, gt):
        if hflip:
            t = t.flip(2)
        if vflip:
            t = t.flip(1)
        if rot90:
            t = t.transpose(1, 2)
        out.append(t.contiguous())
    return out[0], out[1]


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
        if self.opt.get("phase") == "train":
            lq, gt = paired_random_crop_augment(lq, gt, self.opt)
        name = f"synthetic_{index:04d}"
        return _labelled({"lq": lq, "gt": gt, "lq_path": name, "gt_path": name}, self.opt)


def _center_crop(t, size):
    """transforms.center_crop on a CHW tensor"""
    size = (size, size) if isinstance(size, int) else tuple(size)
    _, h, w = t.shape
    top, left = (h - size[0]) // 2, (w - size[1]) // 2
    return t[:, top:top + size[0], left:left + size[1]]


@DATASET_REGISTRY.register()
class PairedImageDataset(tdata.Dataset):
    """Paired folders with identical file names (reference basicsr/data/paired_image_dataset.py:25-192, folder mode,
    PIL instead of cv2): train phase = paired random crop + flips / rotation, other phases = optional centre crop."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.gt_root, self.lq_root = opt["dataroot_gt"], opt.get("dataroot_lq") or opt["dataroot_gt"]
        self.names = sorted(f for f in scandir(self.gt_root) if f.endswith(_IMG_EXT))

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        name = self.names[index]
        gt = _read_rgb(osp.join(self.gt_root, name))
        lq = _read_rgb(osp.join(self.lq_root, name))
        if self.opt.get("phase") == "train":
            lq, gt = paired_random_crop_augment(lq, gt, self.opt)
        elif self.opt.get("center_crop") is not None:
            lq, gt = _center_crop(lq, self.opt["center_crop"]), _center_crop(gt, self.opt["center_crop"])
        return _labelled({"lq": lq, "gt": gt, "lq_path": osp.join(self.lq_root, name), "gt_path": osp.join(self.gt_root, name)}, self.opt)


@DATASET_REGISTRY.register()
class PairedImageDenoiseDataset(tdata.Dataset):
    """GT folder + on-the-fly Gaussian noise (reference basicsr/data/paired_image_dataset.py:276-421).  The noise reproduces
    the reference's draw exactly (:388-402): ``sigma`` from ``sigma_type`` constant / random / choice (Python's ``random``),
    ``np.random.seed(index)`` in the train phase and ``seed(0)`` otherwise, ``normal(0, sigma / 255, (H, W, 3))`` on the
    **HWC**, RGB-ordered float32 image AFTER the crop / augmentation, added in float32."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.gt_root = opt["dataroot_gt"]
        self.sigma_type = opt.get("sigma_type", "constant")
        self.sigma_range = opt["sigma_range"]
        assert self.sigma_type in ["constant", "random", "choice"]
        self.names = sorted(f for f in scandir(self.gt_root) if f.endswith(_IMG_EXT))

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        import random

        path = osp.join(self.gt_root, self.names[index])
        gt = _read_rgb(path)
        if self.opt.get("phase") == "train":
            _, gt = paired_random_crop_augment(gt, gt, self.opt)
        elif self.opt.get("center_crop") is not None:
            gt = _center_crop(gt, self.opt["center_crop"])
        if self.sigma_type == "constant":
            sigma = self.sigma_range
        elif self.sigma_type == "random":
            sigma = random.uniform(self.sigma_range[0], self.sigma_range[1])
        else:
            sigma = random.choice(self.sigma_range)
        rs = np.random.RandomState(index if self.opt.get("phase") == "train" else 0)   # == np.random.seed(...); np.random.normal
        hwc = gt.permute(1, 2, 0).contiguous().numpy().copy()
        hwc += rs.normal(0, sigma / 255.0, hwc.shape)   # float64 noise accumulated into the float32 image, as `img_lq += ...`
        lq = torch.from_numpy(hwc.transpose(2, 0, 1)).float().contiguous()
        return _labelled({"lq": lq, "gt": gt.contiguous(), "lq_path": path, "gt_path": path}, self.opt)


DATASET_REGISTRY._obj_map["PairedImageDehazeDataset"] = PairedImageDataset


def build_dataset(dataset_opt):
    dataset_opt = deepcopy(dataset_opt)
    logger = get_root_logger()
    root = dataset_opt.get("dataroot_gt")
    if dataset_opt["type"] != "SyntheticPairedDataset" and (root is None or not osp.isdir(root)):
        if dataset_opt.get("phase") == "train" and not dataset_opt.get("allow_synthetic", False):
            raise FileNotFoundError(f"Dataset root {root} of the training set [{dataset_opt.get('name')}] does not exist "
                                    "(set `allow_synthetic: true` in the dataset options to train on seeded synthetic pairs)")
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
    if phase == "train":  # reference basicsr/data/__init__.py:64-92: per-GPU batch, sampler-driven order, drop_last
        workers = dataset_opt.get("num_worker_per_gpu", 0) * (1 if dist else max(1, num_gpu))
        batch = dataset_opt["batch_size_per_gpu"] * (1 if dist else max(1, num_gpu))
        rank = torch.distributed.get_rank() if dist else 0

        def worker_init(worker_id):
            import random

            s = (seed or 0) + workers * rank + worker_id
            np.random.seed(s)
            random.seed(s)

        return tdata.DataLoader(dataset, batch_size=batch, shuffle=sampler is None, sampler=sampler, num_workers=workers,
                                drop_last=True, worker_init_fn=worker_init if seed is not None else None,
                                pin_memory=dataset_opt.get("pin_memory", False), persistent_workers=False)
    raise ValueError(f"Wrong dataset phase: {phase}. Supported ones are 'train', 'val' and 'test'.")
