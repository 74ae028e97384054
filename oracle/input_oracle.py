"""CPU oracle for the paired input transform.  TEST INFRASTRUCTURE ONLY.

Restates, in plain fp32 torch, what the reference's ``PairedDataset.__getitem__`` does to a decoded image pair when
``use_augmentation`` is on (src/data/paired_data_module.py:170-199): ``TF.crop(img, i, j, h, w)`` on both images,
optional ``TF.hflip`` / ``TF.vflip`` on both, ``TF.to_tensor`` (uint8 HWC -> float32 CHW, divided by 255) and
``Normalize(mean=0.5, std=0.5)``.

The ``use_augmentation=False`` branch (:200-211) resizes instead: ``TF.resize(img, (S, S))`` on a PIL image, i.e.
``PIL.Image.resize((S, S), Image.BILINEAR)`` -- Pillow's two-pass (horizontal, then vertical) area-aware triangle
filter in 22-bit fixed point with a uint8 intermediate (Pillow 12.2.0, src/libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; third-party source not
in /root/reference, algorithm restated from the published implementation) -- then ``to_tensor`` and ``Normalize``.

Parity status: PINNED through Pillow.  torchvision (the wrapper the reference calls) is not installed, but the
functions it dispatches to for PIL images are Pillow's own -- ``Image.crop``, ``Image.transpose(FLIP_LEFT_RIGHT /
FLIP_TOP_BOTTOM)``, ``Image.resize(..., BILINEAR)`` -- and Pillow is importable here and on the GPU box:
``tests/golden/make_golden.py`` writes ``input_pipeline.npz`` with Pillow's outputs and
``tests/test_input_pipeline.py`` checks this restatement against them bit for bit.  ``to_tensor`` / ``Normalize``
are the documented formulas (permute, /255, (x - 0.5) / 0.5).
"""
import math

import numpy as np
import torch


def paired_transform(src_u8: torch.Tensor, tgt_u8: torch.Tensor, params: torch.Tensor, size: int):
    """src/tgt: uint8 [B,H,W,3]; params int [B,4] = (top, left, hflip, vflip) -> two float32 [B,3,size,size]."""
    outs = []
    for imgs in (src_u8, tgt_u8):
        res = []
        for n in range(imgs.shape[0]):
            top, left, hf, vf = (int(v) for v in params[n])
            crop = imgs[n, top:top + size, left:left + size, :]
            if hf:
                crop = crop.flip(1)
            if vf:
                crop = crop.flip(0)
            t = crop.permute(2, 0, 1).to(torch.float32).div(255)
            res.append((t - 0.5) / 0.5)
        outs.append(torch.stack(res))
    return outs[0], outs[1]


# ---------------------------------------------------------------------------------------------------------------
# PIL.Image.resize(..., BILINEAR) on 8-bit images
# ---------------------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """(bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize]) exactly as Pillow's precompute_coeffs +
    normalize_coeffs_8bpc build them (double arithmetic, triangle filter of support 1 scaled by max(scale, 1))."""
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
        w = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = sum(w)                                  # Pillow adds them in this order too
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            v = k * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis_u8(img: np.ndarray, axis: int, out_size: int) -> np.ndarray:
    """One Pillow pass along ``axis`` of a uint8 [H, W, C] image: sum of uint8 * int32 coefficient, + half, >> 22,
    clipped to 0..255."""
    in_size = img.shape[axis]
    bounds, kk = pil_bilinear_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(cnt):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear_u8(img_u8: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [out_h, out_w, C]; horizontal pass first (skipped when the width is unchanged), then
    vertical, with the uint8 intermediate Pillow keeps between them."""
    h, w = img_u8.shape[:2]
    cur = img_u8
    if out_w != w:
        cur = _resample_axis_u8(cur, 1, out_w)
    if out_h != h:
        cur = _resample_axis_u8(cur, 0, out_h)
    return cur


def resize_transform(img_u8: torch.Tensor, size: int) -> torch.Tensor:
    """uint8 [B, H, W, 3] -> float32 [B, 3, size, size]: resize, to_tensor, Normalize(0.5, 0.5)."""
    outs = []
    for n in range(img_u8.shape[0]):
        r = torch.from_numpy(pil_resize_bilinear_u8(img_u8[n].numpy(), size, size))
        outs.append((r.permute(2, 0, 1).to(torch.float32).div(255) - 0.5) / 0.5)
    return torch.stack(outs)
