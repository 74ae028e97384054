// Paired tile preparation on the GPU: crop + flip (shared parameters for source and target) + to_tensor + Normalize.
//
// Replaces the per-sample CPU transform chain of the reference's PairedDataset.__getitem__
// (src/data/paired_data_module.py:170-199): RandomCrop.get_params -> TF.crop on both images with the SAME
// (i, j, h, w), shared horizontal / vertical flips, TF.to_tensor (uint8 HWC -> float CHW, x/255) and
// Normalize(mean 0.5, std 0.5) -> (x/255 - 0.5)/0.5.  Decoding (cv2/PIL) stays on the host; this kernel takes
// the decoded uint8 HWC RGB images resident in HBM and writes the NCHW fp32 tensors the model boundary expects.
// One thread per output pixel: 3-byte reads are contiguous across lanes, the three plane writes coalesced.
// Bit-exact with the fp32 expression above (IEEE division, same operation order).
#include "common.h"

namespace {

__global__ void paired_prepare_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ tgt,
                                      const int* __restrict__ params, float* __restrict__ out_src,
                                      float* __restrict__ out_tgt, int B, int Hs, int Ws, int S) {
  const long total = (long)B * S * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S);
    const int y = (int)((i / S) % S);
    const int n = (int)(i / ((long)S * S));
    const int top = params[n * 4 + 0], left = params[n * 4 + 1];
    const int hflip = params[n * 4 + 2], vflip = params[n * 4 + 3];
    const int sy = top + (vflip ? S - 1 - y : y);
    const int sx = left + (hflip ? S - 1 - x : x);
    const long ip = (((long)n * Hs + sy) * Ws + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long op = (((long)n * 3 + c) * S + y) * S + x;
      out_src[op] = ((float)src[ip + c] / 255.0f - 0.5f) / 0.5f;
      out_tgt[op] = ((float)tgt[ip + c] / 255.0f - 0.5f) / 0.5f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The no-augmentation branch (paired_data_module.py:200-211): TF.resize(img, (S, S)) on a PIL image =
// PIL.Image.resize((S, S), BILINEAR), then to_tensor + Normalize.  Pillow resamples in two passes -- along the
// width, then along the height -- with per-output-pixel windows (bounds) of 22-bit fixed-point coefficients and a
// uint8 image in between; both passes are  clip8((2^21 + sum_k u8[xmin+k] * kk[k]) >> 22).  The coefficient tables
// are built on the host in double precision exactly as Pillow builds them (stain2stain_amd/data.py), so the result
// is bit-identical.  One thread per output pixel (3 channels), byte reads contiguous across lanes.
// ---------------------------------------------------------------------------------------------
constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PIL_PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// in: uint8 [B][H][Win][3] -> out: uint8 [B][H][Wout][3]
__global__ void pil_resize_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                    const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int B,
                                    int H, int Win, int Wout) {
  const int total = B * H * Wout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / Wout, xx = i - row * Wout;
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int* k = kk + xx * ksize;
    const unsigned char* src = in + ((long)row * Win + xmin) * 3;
    int a0 = 1 << (PIL_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < cnt; ++x) {
      const int w = k[x];
      a0 += src[3 * x + 0] * w; a1 += src[3 * x + 1] * w; a2 += src[3 * x + 2] * w;
    }
    unsigned char* dst = out + (long)i * 3;
    dst[0] = (unsigned char)clip8(a0); dst[1] = (unsigned char)clip8(a1); dst[2] = (unsigned char)clip8(a2);
  }
}

// in: uint8 [B][Hin][W][3] -> out_u8 (optional) uint8 [B][Hout][W][3] and / or out_f (optional) float [B][3][Hout][W]
// = (v/255 - 0.5)/0.5.  bounds == nullptr: no vertical resampling (Hin == Hout), conversion only.
__global__ void pil_resize_v_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out_u8,
                                    float* __restrict__ out_f, const int* __restrict__ bounds,
                                    const int* __restrict__ kk, int ksize, int B, int Hin, int Hout, int W) {
  const int total = B * Hout * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int x = i % W, t = i / W, yy = t % Hout, n = t / Hout;
    int v0, v1, v2;
    if (bounds) {
      const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
      const int* k = kk + yy * ksize;
      const unsigned char* src = in + (((long)n * Hin + ymin) * W + x) * 3;
      int a0 = 1 << (PIL_PRECISION_BITS - 1), a1 = a0, a2 = a0;
      for (int y = 0; y < cnt; ++y) {
        const int w = k[y];
        const unsigned char* p = src + (long)y * W * 3;
        a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
      }
      v0 = clip8(a0); v1 = clip8(a1); v2 = clip8(a2);
    } else {
      const unsigned char* p = in + (((long)n * Hin + yy) * W + x) * 3;
      v0 = p[0]; v1 = p[1]; v2 = p[2];
    }
    if (out_u8) {
      unsigned char* d = out_u8 + (long)i * 3;
      d[0] = (unsigned char)v0; d[1] = (unsigned char)v1; d[2] = (unsigned char)v2;
    }
    if (out_f) {
      const long plane = (long)Hout * W, o = (long)n * 3 * plane + (long)yy * W + x;
      out_f[o] = ((float)v0 / 255.0f - 0.5f) / 0.5f;
      out_f[o + plane] = ((float)v1 / 255.0f - 0.5f) / 0.5f;
      out_f[o + 2 * plane] = ((float)v2 / 255.0f - 0.5f) / 0.5f;
    }
  }
}

}  // namespace

// PIL.Image.resize(BILINEAR) + to_tensor + Normalize(0.5, 0.5) of a batch of decoded RGB images.
//   src: uint8 [B][Hs][Ws][3]; tmp: uint8 [B][Hs][Wo][3] scratch (unused when Ws == Wo);
//   bounds_h / kk_h: int32 [Wo][2] / [Wo][ksize_h] (NULL when Ws == Wo); bounds_v / kk_v likewise for the height;
//   out_u8 (optional): uint8 [B][Ho][Wo][3], the resized image; out_f (optional): float [B][3][Ho][Wo].
extern "C" int s2s_pil_resize_bilinear_normalize(const void* src_u8, void* tmp_u8, const int* bounds_h,
                                                 const int* kk_h, int ksize_h, const int* bounds_v, const int* kk_v,
                                                 int ksize_v, void* out_u8, float* out_f, int B, int Hs, int Ws,
                                                 int Ho, int Wo, void* stream) {
  if (!src_u8 || (!out_u8 && !out_f)) return S2S_ERR_NULL;
  if (B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return S2S_ERR_SHAPE;
  const bool need_h = Ws != Wo, need_v = Hs != Ho;
  if (need_h && (!bounds_h || !kk_h || ksize_h <= 0 || !tmp_u8)) return S2S_ERR_NULL;
  if (need_v && (!bounds_v || !kk_v || ksize_v <= 0)) return S2S_ERR_NULL;
  if ((long)B * Hs * (Ws > Wo ? Ws : Wo) >= (1L << 31) - (1L << 22) || (long)B * Ho * Wo >= (1L << 31) - (1L << 22))
    return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const unsigned char* cur = (const unsigned char*)src_u8;
  if (need_h) {
    const int total = B * Hs * Wo;
    int grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(pil_resize_h_kernel, dim3(grid), dim3(256), 0, s, cur, (unsigned char*)tmp_u8, bounds_h, kk_h,
                       ksize_h, B, Hs, Ws, Wo);
    cur = (const unsigned char*)tmp_u8;
  }
  const int total = B * Ho * Wo;
  int grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(pil_resize_v_kernel, dim3(grid), dim3(256), 0, s, cur, (unsigned char*)out_u8, out_f,
                     need_v ? bounds_v : nullptr, kk_v, ksize_v, B, Hs, Ho, Wo);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// src/tgt: uint8 [B][Hs][Ws][3] (decoded RGB, HWC); params: int32 [B][4] = {top, left, hflip, vflip};
// out_*: float [B][3][S][S].  Crop windows must lie inside the images (validated on the host by the caller's
// parameter generator; the kernel does not clamp).
extern "C" int s2s_paired_crop_flip_normalize(const void* src_u8, const void* tgt_u8, const int* params,
                                              float* out_src, float* out_tgt, int B, int Hs, int Ws, int S,
                                              void* stream) {
  if (!src_u8 || !tgt_u8 || !params || !out_src || !out_tgt) return S2S_ERR_NULL;
  if (B <= 0 || S <= 0 || Hs < S || Ws < S) return S2S_ERR_SHAPE;
  const long total = (long)B * S * S;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(paired_prepare_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src_u8, (const unsigned char*)tgt_u8, params, out_src, out_tgt, B, Hs, Ws,
                     S);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
