"""METRIC_REGISTRY: cv2-free PSNR / SSIM with the reference's interface (basicsr/metrics/psnr_ssim.py:11-75 ``calculate_psnr``,
:113-183 ``calculate_ssim``, ``_ssim`` :483-512; ``reorder_image`` / ``to_y_channel`` of metric_util.py):

    calculate_psnr(img, img2, crop_border, input_order="BCHW", test_y_channel=False, image_range=255)

``img`` / ``img2`` are float arrays in [0, 1] -- a batch (B,C,H,W) / (B,H,W,C), one image (C,H,W), or a 2-D map -- which are
scaled by ``image_range`` and ROUNDED to uint8 (uint16 for other ranges; ``image_range=1`` skips the quantisation), cropped,
optionally reduced to BT.601 luma, compared in float64 and averaged over the batch.  The reference's RGB->BGR flip
(cv2.cvtColor) only matters for the luma weights, which are applied here in the same B, G, R order.

Extension kept for callers that already hold quantised images: ``input_order`` "HWC" / "CHW" takes ONE image whose values are
already in [0, image_range] (no scaling, no rounding).

PSNR is pinned against the reference function itself (tests/golden/metrics.npz, oracle/make_golden.py::gen_metrics: the reference's
``calculate_psnr`` only needs ``cv2.cvtColor``'s channel flip, which a three-line module object provides).

SSIM stays RESTATEMENT-ONLY, deliberately: the arithmetic of the reference's ``_ssim`` (:483-512) lives in ``cv2.getGaussianKernel``
and ``cv2.filter2D``, and cv2 is not installed in the build image.  Running the reference function would therefore mean running it
on stand-ins for those two OpenCV routines written here -- the result would pin this file against its own author's reading of
OpenCV, not against the reference, so no such fixture is committed.  What is checked instead: ``calculate_ssim`` against an
independently written scipy implementation of the published formula (11 x 11 Gaussian, sigma 1.5, "valid" window, C1 = (0.01 L)^2,
C2 = (0.03 L)^2, float64; tests/test_plumbing_cpu.py::test_metrics_match_independent_implementations), and the acceptance gate of
``north_star`` is stated in PSNR (<= 0.01 dB), which IS pinned."""
from copy import deepcopy

import numpy as np

from basicsr.utils.registry import METRIC_REGISTRY

__all__ = ["calculate_metric", "calculate_psnr", "calculate_ssim"]


def _reorder(img, input_order):
    """metric_util.reorder_image: anything -> (B, H, W, C)"""
    if input_order not in ("BHWC", "BCHW"):
        raise ValueError(f"Wrong input_order {input_order}. Supported input_orders are 'HWC' and 'CHW'")
    if img.ndim == 2:
        img = img[None, ..., None]
    if input_order == "BCHW":
        if img.ndim == 3:
            img = img.transpose(1, 2, 0)[None, ...]
        elif img.ndim == 4:
            img = img.transpose(0, 2, 3, 1)
    elif img.ndim == 3:
        img = img[None, ...]
    return img


def _to_y(img_rgb, image_range):
    """metric_util.to_y_channel on the BGR-flipped image (color_util.bgr2ycbcr, y_only): float32 in, float32 out, no rounding"""
    x = img_rgb[..., ::-1].astype(np.float32) / image_range
    if x.dtype != np.float32:
        x = x.astype(np.float32)
    y = (np.dot(x, [24.966, 128.553, 65.481]) + 16.0) / 255.0
    return y.astype(np.float32)[..., None] * image_range


def _images(img, img2, crop_border, input_order, test_y_channel, image_range):
    """yields the float64 (H, W, C) pairs the metric is evaluated on"""
    img, img2 = np.asarray(img), np.asarray(img2)
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    legacy = input_order in ("HWC", "CHW")
    if legacy:
        if input_order == "CHW":
            img, img2 = img.transpose(1, 2, 0), img2.transpose(1, 2, 0)
        if img.ndim == 2:
            img, img2 = img[..., None], img2[..., None]
        batch = [(img, img2)]
    else:
        a, b = _reorder(img, input_order), _reorder(img2, input_order)
        batch = [(a[i], b[i]) for i in range(a.shape[0])]
    dtype = np.uint8 if image_range == 255 else np.uint16
    for x, y in batch:
        if not legacy and image_range != 1:
            x = (x * float(image_range)).round().astype(dtype)
            y = (y * float(image_range)).round().astype(dtype)
        if crop_border != 0:
            x = x[crop_border:-crop_border, crop_border:-crop_border, ...]
            y = y[crop_border:-crop_border, crop_border:-crop_border, ...]
        if test_y_channel and x.shape[-1] == y.shape[-1] == 3:
            x, y = _to_y(x, image_range), _to_y(y, image_range)
        yield x.astype(np.float64), y.astype(np.float64)


@METRIC_REGISTRY.register()
def calculate_psnr(img, img2, crop_border, input_order="BCHW", test_y_channel=False, image_range=255, **kwargs):
    psnrs = []
    for x, y in _images(img, img2, crop_border, input_order, test_y_channel, image_range):
        mse = np.mean((x - y) ** 2)
        if mse == 0:
            return float("inf")
        psnrs.append(10.0 * np.log10(image_range * image_range / mse))
    return float(np.array(psnrs).mean())


def _gauss_kernel():
    """cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1"""
    x = np.arange(11, dtype=np.float64) - 5.0
    g = np.exp(-(x ** 2) / (2 * 1.5 ** 2))
    return g / g.sum()


def _filter_valid(img, k):
    """the 11x11 outer-product window as two 1-D 'valid' passes (cv2.filter2D then [5:-5, 5:-5] in the reference)"""
    h = np.apply_along_axis(lambda r: np.convolve(r, k, mode="valid"), 1, img)
    return np.apply_along_axis(lambda c: np.convolve(c, k, mode="valid"), 0, h)


def _ssim(img, img2, image_range=255):
    c1, c2 = (0.01 * image_range) ** 2, (0.03 * image_range) ** 2
    k = _gauss_kernel()
    mu1, mu2 = _filter_valid(img, k), _filter_valid(img2, k)
    mu1_sq, mu2_sq, mu12 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _filter_valid(img ** 2, k) - mu1_sq
    s2 = _filter_valid(img2 ** 2, k) - mu2_sq
    s12 = _filter_valid(img * img2, k) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    return (((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map).mean()


@METRIC_REGISTRY.register()
def calculate_ssim(img, img2, crop_border, input_order="BCHW", test_y_channel=False, image_range=255, **kwargs):
    ssims = []
    for x, y in _images(img, img2, crop_border, input_order, test_y_channel, image_range):
        for j in range(x.shape[2]):
            ssims.append(_ssim(x[..., j], y[..., j], image_range))
    return float(np.array(ssims).mean())


def calculate_metric(data, opt):
    opt = deepcopy(opt)
    return METRIC_REGISTRY.get(opt.pop("type"))(**data, **opt)
