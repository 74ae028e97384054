"""GPU side of the paired-tile input pipeline.

Mirrors the transform chain of the reference's ``PairedDataset.__getitem__``
(src/data/paired_data_module.py:149-223) from the decoded images onward: one random crop window and one pair of
flip decisions per sample, applied identically to source and target, then ``to_tensor`` and
``Normalize(0.5, 0.5)``.  File decoding (cv2 / PIL) stays on the CPU and is out of scope.
"""
from __future__ import annotations

import random
from typing import Optional, Tuple

import torch

import math
from functools import lru_cache

from . import _native, ops


def sample_crop_flip_params(batch: int, src_hw: Tuple[int, int], size: int, rng: Optional[random.Random] = None
                            ) -> torch.Tensor:
    """int32 [B, 4] = (top, left, hflip, vflip), drawn like the reference does: ``RandomCrop.get_params`` picks
    top ~ U{0..H-size}, left ~ U{0..W-size}; each flip happens when ``random.random() > 0.5``
    (paired_data_module.py:172-191)."""
    rng = rng or random
    h, w = src_hw
    if h < size or w < size:
        raise ValueError(f"crop size {size} exceeds the image size {src_hw}")
    rows = []
    for _ in range(batch):
        top = rng.randint(0, h - size)
        left = rng.randint(0, w - size)
        rows.append([top, left, int(rng.random() > 0.5), int(rng.random() > 0.5)])
    return torch.tensor(rows, dtype=torch.int32)


def paired_crop_flip_normalize(src_u8: torch.Tensor, tgt_u8: torch.Tensor, params: torch.Tensor, size: int):
    """src_u8/tgt_u8: uint8 [B, H, W, 3] on the GPU; params: int32 [B, 4]; returns two fp32 [B, 3, size, size]."""
    if src_u8.dtype != torch.uint8 or tgt_u8.dtype != torch.uint8 or src_u8.shape != tgt_u8.shape:
        raise RuntimeError("stain2stain_amd: expected two uint8 [B,H,W,3] tensors of the same shape")
    if not src_u8.is_cuda or src_u8.dim() != 4 or src_u8.shape[3] != 3:
        raise RuntimeError("stain2stain_amd: expected GPU tensors in HWC RGB layout")
    B, H, W, _ = src_u8.shape
    p = params.to(device=src_u8.device, dtype=torch.int32).contiguous()
    pc = params.cpu()
    if p.shape != (B, 4) or int(pc[:, 0].min()) < 0 or int(pc[:, 1].min()) < 0 or \
            int(pc[:, 0].max()) + size > H or int(pc[:, 1].max()) + size > W:
        raise RuntimeError("stain2stain_amd: crop window outside the image")
    out_s = torch.empty((B, 3, size, size), dtype=torch.float32, device=src_u8.device)
    out_t = torch.empty_like(out_s)
    rc = _native.lib().s2s_paired_crop_flip_normalize(src_u8.contiguous().data_ptr(), tgt_u8.contiguous().data_ptr(),
                                                      p.data_ptr(), out_s.data_ptr(), out_t.data_ptr(), B, H, W, size,
                                                      ops._stream())
    _native.check(rc, "paired_crop_flip_normalize")
    return out_s, out_t


# ----------------------------------------------------------------------------------------------------------------
# the use_augmentation=False branch: TF.resize on a PIL image + to_tensor + Normalize
# ----------------------------------------------------------------------------------------------------------------
_PIL_PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=64)
def pil_bilinear_tables(in_size: int, out_size: int):
    """Pillow's resampling windows for one axis: (bounds int32 [out, 2], kk int32 [out, ksize], ksize), built as
    Resample.c builds them (precompute_coeffs in double precision with the triangle filter of support
    max(in/out, 1), normalize_coeffs_8bpc to 22-bit fixed point)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        row = [0] * ksize
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            v = k * (1 << _PIL_PRECISION_BITS)
            row[x] = int(-0.5 + v) if k < 0 else int(0.5 + v)
        bounds.append([xmin, xmax])
        kk.append(row)
    return (torch.tensor(bounds, dtype=torch.int32), torch.tensor(kk, dtype=torch.int32), ksize)


def resize_normalize(img_u8: torch.Tensor, size: int, want_u8: bool = False):
    """uint8 [B, H, W, 3] on the GPU -> float32 [B, 3, size, size] = Normalize(to_tensor(PIL resize BILINEAR));
    with ``want_u8`` also the resized uint8 [B, size, size, 3] image."""
    if img_u8.dtype != torch.uint8 or not img_u8.is_cuda or img_u8.dim() != 4 or img_u8.shape[3] != 3:
        raise RuntimeError("stain2stain_amd: expected a uint8 [B,H,W,3] tensor on the GPU")
    B, H, W, _ = img_u8.shape
    dev = img_u8.device
    img_u8 = img_u8.contiguous()
    bh = kh = bv = kv = None
    ksh = ksv = 0
    tmp = None
    if W != size:
        bh, kh, ksh = pil_bilinear_tables(W, size)
        bh, kh = bh.to(dev), kh.to(dev)
        tmp = torch.empty((B, H, size, 3), dtype=torch.uint8, device=dev)
    if H != size:
        bv, kv, ksv = pil_bilinear_tables(H, size)
        bv, kv = bv.to(dev), kv.to(dev)
    out_f = torch.empty((B, 3, size, size), dtype=torch.float32, device=dev)
    out_u8 = torch.empty((B, size, size, 3), dtype=torch.uint8, device=dev) if want_u8 else None
    ptr = lambda t: 0 if t is None else t.data_ptr()      # noqa: E731
    rc = _native.lib().s2s_pil_resize_bilinear_normalize(img_u8.data_ptr(), ptr(tmp), ptr(bh), ptr(kh), ksh, ptr(bv),
                                                         ptr(kv), ksv, ptr(out_u8), out_f.data_ptr(), B, H, W, size,
                                                         size, ops._stream())
    _native.check(rc, "pil_resize_bilinear_normalize")
    return (out_f, out_u8) if want_u8 else out_f
