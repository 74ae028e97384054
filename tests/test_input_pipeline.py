"""Row f1: paired crop/flip/normalise.  Bit-exact against the CPU oracle (integer indexing + one fp32 expression)."""
import random

import pytest
import torch


def test_param_sampler_ranges():
    from stain2stain_amd.data import sample_crop_flip_params
    p = sample_crop_flip_params(64, (300, 280), 256, random.Random(1984))
    assert p.shape == (64, 4) and p.dtype == torch.int32
    assert int(p[:, 0].min()) >= 0 and int(p[:, 0].max()) <= 44 and int(p[:, 1].max()) <= 24
    assert set(p[:, 2].tolist()) <= {0, 1} and set(p[:, 3].tolist()) <= {0, 1}
    with pytest.raises(ValueError):
        sample_crop_flip_params(1, (100, 100), 256)


def test_oracle_matches_elementary_definition():
    from oracle.input_oracle import paired_transform
    img = torch.arange(2 * 5 * 6 * 3, dtype=torch.uint8).reshape(2, 5, 6, 3)
    a, b = paired_transform(img, img, torch.tensor([[1, 2, 0, 0], [0, 1, 1, 1]]), 3)
    assert torch.equal(a, b)
    f = lambda v: (v.to(torch.float32) / 255 - 0.5) / 0.5
    assert a[0, 1, 0, 0] == f(img[0, 1, 2, 1])
    assert a[1, 2, 0, 0] == f(img[1, 2, 3, 2])      # both flips: (y,x) <- (2-y, 2-x) + offset


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 70, 90, 64), (2, 256, 256, 256), (16, 512, 512, 256)])
def test_gpu_transform_is_bit_exact(shape):
    from oracle.input_oracle import paired_transform
    from stain2stain_amd.data import paired_crop_flip_normalize, sample_crop_flip_params
    B, H, W, S = shape
    g = torch.Generator().manual_seed(5)
    src = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    tgt = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    params = sample_crop_flip_params(B, (H, W), S, random.Random(7))
    ref_s, ref_t = paired_transform(src, tgt, params, S)
    out_s, out_t = paired_crop_flip_normalize(src.cuda(), tgt.cuda(), params, S)
    assert torch.equal(out_s.cpu(), ref_s) and torch.equal(out_t.cpu(), ref_t)
    assert float(out_s.min()) >= -1.0 and float(out_s.max()) <= 1.0
    with pytest.raises(RuntimeError):
        bad = params.clone(); bad[0, 0] = H
        paired_crop_flip_normalize(src.cuda(), tgt.cuda(), bad, S)
