"""METRIC_REGISTRY: cv2-free PSNR / SSIM as the YAMLs use them (reference basicsr/metrics/psnr_ssim.py:11-75,
:113-183, _ssim :483-512): uint8-rounded HWC images in [0,255], optional border crop, float64 math."""
from copy import deepcopy

import numpy as np

from basicsr.utils.registry import METRIC_REGISTRY

__all__ = ["calculate_metric", "calculate_psnr", "calculate_ssim"]


def _to_y(img):
    """BT.601 luma of a BGR/RGB-agnostic HWC image in [0,255] is order dependent; inputs here are RGB."""
    img = img.astype(np.float32) / 255.0
    y = np.dot(img, [65.481, 128.553, 24.966]) + 16.0
    return y[..., None]


def _prep(img, img2, crop_border, input_order, test_y_channel):
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    if input_order not in ("HWC", "CHW"):
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    if input_order == "CHW":
        img, img2 = img.transpose(1, 2, 0), img2.transpose(1, 2, 0)
    if img.ndim == 2:
        img, img2 = img[..., None], img2[..., None]
    if crop_border != 0:
        img = img[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img, img2 = _to_y(img), _to_y(img2)
    return img.astype(np.float64), img2.astype(np.float64)


@METRIC_REGISTRY.register()
def calculate_psnr(img, img2, crop_border, input_order="HWC", test_y_channel=False, **kwargs):
    img, img2 = _prep(img, img2, crop_border, input_order, test_y_channel)
    mse = np.mean((img - img2) ** 2)
    if mse == 0:
        return float("inf")
    return 10.0 * np.log10(255.0 * 255.0 / mse)


def _gauss_kernel():
    x = np.arange(11, dtype=np.float64) - 5.0
    g = np.exp(-(x ** 2) / (2 * 1.5 ** 2))
    return g / g.sum()


def _filter_valid(img, k):
    """separable 11-tap 'valid' correlation (cv2.filter2D then [5:-5, 5:-5] in the reference)."""
    h = np.apply_along_axis(lambda r: np.convolve(r, k, mode="valid"), 1, img)
    return np.apply_along_axis(lambda c: np.convolve(c, k, mode="valid"), 0, h)


def _ssim(img, img2):
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    k = _gauss_kernel()
    mu1, mu2 = _filter_valid(img, k), _filter_valid(img2, k)
    mu1_sq, mu2_sq, mu12 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _filter_valid(img ** 2, k) - mu1_sq
    s2 = _filter_valid(img2 ** 2, k) - mu2_sq
    s12 = _filter_valid(img * img2, k) - mu12
    return (((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))).mean()


@METRIC_REGISTRY.register()
def calculate_ssim(img, img2, crop_border, input_order="HWC", test_y_channel=False, **kwargs):
    img, img2 = _prep(img, img2, crop_border, input_order, test_y_channel)
    return float(np.mean([_ssim(img[..., i], img2[..., i]) for i in range(img.shape[2])]))


def calculate_metric(data, opt):
    opt = deepcopy(opt)
    return METRIC_REGISTRY.get(opt.pop("type"))(**data, **opt)
