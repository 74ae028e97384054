"""CPU oracle for the paired input transform.  TEST INFRASTRUCTURE ONLY.

Restates, in plain fp32 torch, what the reference's ``PairedDataset.__getitem__`` does to a decoded image pair when
``use_augmentation`` is on (src/data/paired_data_module.py:170-199): ``TF.crop(img, i, j, h, w)`` on both images,
optional ``TF.hflip`` / ``TF.vflip`` on both, ``TF.to_tensor`` (uint8 HWC -> float32 CHW, divided by 255) and
``Normalize(mean=0.5, std=0.5)``.

Parity status: UNPINNED by the reference.  The transform functions live in torchvision (third party, not installed
in the build container; the reference pins no test vectors for its data pipeline), so this file restates their
documented behaviour: crop = array slice [i:i+h, j:j+w], hflip / vflip = reversal of the width / height axis,
to_tensor = permute + /255, Normalize = (x - mean) / std.
"""
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
