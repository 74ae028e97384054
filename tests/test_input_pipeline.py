"""Row f1: paired crop / flip / resize / normalise.  The oracle is pinned, bit for bit, by Pillow's own outputs
(tests/golden/input_pipeline.npz: the PIL calls torchvision's TF.crop / hflip / vflip / resize dispatch to); the
GPU kernels are bit-exact against the oracle (integer indexing, 22-bit fixed-point resampling, one fp32 expression)."""
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


# ---- pinned by Pillow -------------------------------------------------------------------------------------------
def _golden():
    import numpy as np
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "input_pipeline.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_oracle_crop_flip_matches_pillow():
    from oracle.input_oracle import paired_transform
    G = _golden()
    img = torch.from_numpy(G["aug/img"])[None]
    for i, prm in enumerate(G["aug/params"]):
        a, _ = paired_transform(img, img, torch.tensor(prm)[None], 64)
        want = torch.from_numpy(G[f"aug/out{i}"]).permute(2, 0, 1).float().div(255).sub(0.5).div(0.5)
        assert torch.equal(a[0], want), i


def test_oracle_resize_matches_pillow_bit_for_bit():
    from oracle.input_oracle import pil_resize_bilinear_u8, resize_transform
    G = _golden()
    i = 0
    while f"resize/in{i}" in G:
        got = pil_resize_bilinear_u8(G[f"resize/in{i}"], 64, 64)
        assert (got == G[f"resize/out{i}"]).all(), i
        i += 1
    assert i == 5
    t = resize_transform(torch.from_numpy(G["resize/in1"])[None], 64)
    want = torch.from_numpy(G["resize/out1"]).permute(2, 0, 1).float().div(255).sub(0.5).div(0.5)
    assert torch.equal(t[0], want)


def test_host_tables_equal_the_oracle_tables():
    import numpy as np
    from oracle.input_oracle import pil_bilinear_coeffs
    from stain2stain_amd.data import pil_bilinear_tables
    for a, b in [(512, 256), (300, 256), (200, 256), (47, 61), (64, 64), (5, 256)]:
        bo, kk = pil_bilinear_coeffs(a, b)
        tb, tk, ks = pil_bilinear_tables(a, b)
        assert ks == kk.shape[1] and np.array_equal(bo, tb.numpy()) and np.array_equal(kk, tk.numpy())


@pytest.mark.gpu
def test_gpu_resize_matches_pillow_golden():
    from stain2stain_amd.data import resize_normalize
    G = _golden()
    i = 0
    while f"resize/in{i}" in G:
        f, u8 = resize_normalize(torch.from_numpy(G[f"resize/in{i}"])[None].cuda(), 64, want_u8=True)
        want = torch.from_numpy(G[f"resize/out{i}"])
        assert torch.equal(u8[0].cpu(), want), i
        assert torch.equal(f[0].cpu(), want.permute(2, 0, 1).float().div(255).sub(0.5).div(0.5)), i
        i += 1


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 512, 512, 256), (3, 300, 280, 256), (2, 200, 190, 256), (1, 256, 300, 256),
                                   (4, 256, 256, 256), (1, 5, 7, 64)])
def test_gpu_resize_is_bit_exact_against_the_oracle(shape):
    """Production sizes: 2x down-sampling (the 512-pixel datasets at image_size 256), ragged down / up-sampling,
    one axis unchanged, nothing to resize (conversion only), tiny sources."""
    from oracle.input_oracle import resize_transform
    from stain2stain_amd.data import resize_normalize
    B, H, W, S = shape
    g = torch.Generator().manual_seed(H + W)
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    got = resize_normalize(img.cuda(), S)
    assert torch.equal(got.cpu(), resize_transform(img, S))
    with pytest.raises(RuntimeError):
        resize_normalize(img, S)                          # host tensor
