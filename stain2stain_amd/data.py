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
