"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image decode that sits in front of the hot path:
``transforms.Resize(input_size)`` + ``transforms.ToTensor()`` applied to ``PIL.Image.fromarray(uint8 HxWx3)``
(/root/reference/mmdyn/pytorch/utils/datasets.py:23-31, 375-385).

The arithmetic lives in third-party dependencies that are not vendored in the reference:

* ``pillow`` (un-pinned in /root/reference/setup.py:18; 12.2.0 is installed here and is what the vectors in
  tests/golden/resize_pil.npz were produced with).  ``Image.resize(size, BILINEAR)`` for 8-bit images is
  libImaging/Resample.c: ``precompute_coeffs`` (triangle filter whose support is scaled by the down-scale
  factor, i.e. anti-aliased), ``normalize_coeffs_8bpc`` (22-bit fixed point) and a horizontal pass followed by a
  vertical pass, each rounding to uint8 (``clip8``).  Restated below in numpy with the same operation order, so
  the result is bit-identical.
* ``torchvision.transforms`` (not installed here): ``Resize(int)`` scales the SHORT side to ``size`` keeping the
  aspect ratio (long side ``int(size * long / short)``) and calls ``img.resize((w, h), BILINEAR)``;
  ``ToTensor`` is HWC uint8 -> CHW float32 ``/ 255``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the full box
    (in0 = 0, in1 = in_size).  Returns (bounds int32 [out][2] = (first tap, tap count), coeffs int32 [out][ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _pass(img, bounds, kk, axis):
    """One resampling pass along ``axis`` (1 = horizontal, 0 = vertical) of a uint8 [H][W][C] image."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += img[lo + t] * int(kk[o, t])
        out[o] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """``Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR)`` for uint8 [H][W][C]."""
    H, W = img.shape[:2]
    if (H, W) == (out_h, out_w):
        return img.copy()
    if W != out_w:
        bx, kx = precompute_coeffs(W, out_w)
        img = _pass(img, bx, kx, 1)
    if H != out_h:
        by, ky = precompute_coeffs(H, out_h)
        img = _pass(img, by, ky, 0)
    return img


def resize_output_size(h, w, size):
    """torchvision Resize(int): short side -> size, long side -> int(size * long / short)."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)     # (out_h, out_w)


def resize_to_tensor(img, size):
    """Resize(size) + ToTensor(): uint8 [H][W][3] -> float32 [3][h][w] in [0, 1]."""
    oh, ow = resize_output_size(img.shape[0], img.shape[1], size)
    r = resize_bilinear_u8(img, oh, ow)
    return np.ascontiguousarray(r.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)
