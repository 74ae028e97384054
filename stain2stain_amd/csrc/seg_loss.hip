// Segmentation loss of the multitask model: Dice + BCE-with-logits on a [B,1,H,W] logit map,
// forward and backward in two passes (the Dice term needs three global sums before its gradient exists).
//
//   DiceLoss.forward                 src/models/conditional_flow_matching_multitask.py:36-53
//       p = sigmoid(z);  dice = 1 - (2*sum(p*g) + smooth) / (sum(p) + sum(g) + smooth)
//   nn.BCEWithLogitsLoss (mean)       :119, used at :191   ->  mean(max(z,0) - z*g + log(1 + exp(-|z|)))
//   compute_segmentation_loss         :174-202             ->  seg = dw*dice + (1-dw)*bce
//
// pass 1: per-workgroup fp64 partials of (sum p*g, sum p, sum g, sum bce) -> finalize writes {seg, dice, bce} and the
//         four sums; pass 2: dz = scale * (dw * d dice/dz + (1-dw) * d bce/dz),
//         d dice/dz = -[2 g (P+G+s) - (2I+s)] / (P+G+s)^2 * p (1-p),   d bce/dz = (p - g) / N.
#include "common.h"

namespace {

constexpr int SEG_BLOCKS = 512;

__global__ __launch_bounds__(256) void seg_loss_reduce_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                              long n, double* __restrict__ part) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float zi = z[i], gi = g[i];
    const float p = 1.f / (1.f + expf(-zi));
    s[0] += (double)(p * gi);
    s[1] += (double)p;
    s[2] += (double)gi;
    s[3] += (double)(fmaxf(zi, 0.f) - zi * gi + log1pf(expf(-fabsf(zi))));
  }
  __shared__ double red[4][256];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) part[(long)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// out[0..2] = seg, dice, bce (float); sums[0..3] = I, P, G, BCEsum (double)
__global__ void seg_loss_finalize_kernel(const double* part, int nblk, long n, float smooth, float dice_w,
                                         float* out, double* sums) {
  if (threadIdx.x >= 4) return;
  double a = 0.0;
  for (int i = 0; i < nblk; ++i) a += part[(long)i * 4 + threadIdx.x];
  sums[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double I = sums[0], P = sums[1], G = sums[2];
    const double dice = 1.0 - (2.0 * I + smooth) / (P + G + smooth);
    const double bce = sums[3] / (double)n;
    out[0] = (float)(dice_w * dice + (1.0 - dice_w) * bce);
    out[1] = (float)dice;
    out[2] = (float)bce;
  }
}

__global__ void seg_loss_bwd_kernel(const float* __restrict__ z, const float* __restrict__ g, long n,
                                    const double* __restrict__ sums, float smooth, float dice_w, float scale,
                                    float* __restrict__ dz) {
  const double I = sums[0], P = sums[1], G = sums[2];
  const double den = P + G + smooth;
  const float a = (float)(2.0 / den), b = (float)((2.0 * I + smooth) / (den * den));
  const float invn = (float)(1.0 / (double)n);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float zi = z[i], gi = g[i];
    const float p = 1.f / (1.f + expf(-zi));
    const float ddice = -(gi * a - b) * p * (1.f - p);
    const float dbce = (p - gi) * invn;
    dz[i] = scale * (dice_w * ddice + (1.f - dice_w) * dbce);
  }
}

// ---------------------------------------------------------------------------------------------
// multiclass variant: softmax Dice (mean over classes) + cross entropy on [B,C,H,W] logits, C <= 8
//   MulticlassDiceLoss.forward   src/models/conditional_flow_matching_multitask_multiclassloss.py:41-83
//   nn.CrossEntropyLoss(ignore_index)                                              :159
//   compute_segmentation_loss                                                      :214-245
// sums layout (double): [0..C) I_c, [8..8+C) P_c, [16..16+C) G_c, [24] CE sum, [25] valid count
// ---------------------------------------------------------------------------------------------
constexpr int MC_MAX = 8, MC_SUMS = 26;

template <int C>
__device__ __forceinline__ void softmax_px(const float* __restrict__ z, long base, long HW, float (&p)[C], float& lse) {
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < C; ++c) { p[c] = z[base + c * HW]; m = fmaxf(m, p[c]); }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { p[c] = expf(p[c] - m); sum += p[c]; }
  const float inv = 1.f / sum;
#pragma unroll
  for (int c = 0; c < C; ++c) p[c] *= inv;
  lse = m + logf(sum);
}

template <int C>
__global__ __launch_bounds__(256) void mc_reduce_kernel(const float* __restrict__ z, const long* __restrict__ tgt,
                                                        long npix, long HW, int ignore_index,
                                                        double* __restrict__ part) {
  double s[3 * C + 2];
#pragma unroll
  for (int k = 0; k < 3 * C + 2; ++k) s[k] = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long n = i / HW, q = i - n * HW;
    const long t = tgt[i];
    float p[C], lse;
    softmax_px<C>(z, n * C * HW + q, HW, p, lse);
    const bool dice_valid = !(ignore_index >= 0 && t == ignore_index);
    if (dice_valid) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float oh = (t == c) ? 1.f : 0.f;
        s[c] += (double)(p[c] * oh);
        s[C + c] += (double)p[c];
        s[2 * C + c] += (double)oh;
      }
    }
    if (t != ignore_index && t >= 0 && t < C) {
      s[3 * C] += (double)(lse - z[n * C * HW + t * HW + q]);
      s[3 * C + 1] += 1.0;
    }
  }
  __shared__ double red[256];
  for (int k = 0; k < 3 * C + 2; ++k) {
    __syncthreads();
    red[threadIdx.x] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const int slot = k < C ? k : k < 2 * C ? 8 + (k - C) : k < 3 * C ? 16 + (k - 2 * C) : 24 + (k - 3 * C);
      part[(long)blockIdx.x * MC_SUMS + slot] = red[0];
    }
  }
}

__global__ void mc_finalize_kernel(const double* part, int nblk, int C, float smooth, float dice_w, float* out,
                                   double* sums) {
  const int k = threadIdx.x;
  if (k < MC_SUMS) {
    const int c = k & 7;
    double a = 0.0;
    if (k >= 24 || c < C)
      for (int i = 0; i < nblk; ++i) a += part[(long)i * MC_SUMS + k];
    sums[k] = a;
  }
  __syncthreads();
  if (k == 0) {
    double md = 0.0;
    for (int c = 0; c < C; ++c) md += (2.0 * sums[c] + smooth) / (sums[8 + c] + sums[16 + c] + smooth);
    const double dice = 1.0 - md / C;
    const double ce = sums[24] / sums[25];          // 0/0 = nan when every pixel is ignored, like torch
    out[0] = (float)(dice_w * dice + (1.0 - dice_w) * ce);
    out[1] = (float)dice;
    out[2] = (float)ce;
  }
}

template <int C>
__global__ __launch_bounds__(256) void mc_bwd_kernel(const float* __restrict__ z, const long* __restrict__ tgt,
                                                     long npix, long HW, int ignore_index,
                                                     const double* __restrict__ sums, float smooth, float dice_w,
                                                     float scale, float* __restrict__ dz) {
  float ka[C], kb[C];            // d(1 - mean dice)/dp_c = -(ka[c] * onehot - kb[c])
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const double den = sums[8 + c] + sums[16 + c] + smooth;
    ka[c] = (float)(2.0 / den / C);
    kb[c] = (float)((2.0 * sums[c] + smooth) / (den * den) / C);
  }
  const float invn = (float)(1.0 / sums[25]);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long n = i / HW, q = i - n * HW;
    const long t = tgt[i];
    float p[C], lse;
    const long base = n * C * HW + q;
    softmax_px<C>(z, base, HW, p, lse);
    const bool dice_valid = !(ignore_index >= 0 && t == ignore_index);
    const bool ce_valid = t != ignore_index && t >= 0 && t < C;
    float a[C], dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      a[c] = dice_valid ? -(ka[c] * ((t == c) ? 1.f : 0.f) - kb[c]) : 0.f;
      dot = fmaf(a[c], p[c], dot);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float ddice = p[c] * (a[c] - dot);
      const float dce = ce_valid ? (p[c] - ((t == c) ? 1.f : 0.f)) * invn : 0.f;
      dz[base + c * HW] = scale * (dice_w * ddice + (1.f - dice_w) * dce);
    }
  }
}

template <int C>
void mc_launch(const float* z, const long* t, float* dz, float* out, double* work, long npix, long HW, int ignore,
               float smooth, float dw, float scale, hipStream_t s) {
  long nb = (npix + 255) / 256;
  if (nb > SEG_BLOCKS) nb = SEG_BLOCKS;
  double* sums = work + (long)SEG_BLOCKS * MC_SUMS;
  hipLaunchKernelGGL(mc_reduce_kernel<C>, dim3((int)nb), dim3(256), 0, s, z, t, npix, HW, ignore, work);
  hipLaunchKernelGGL(mc_finalize_kernel, dim3(1), dim3(64), 0, s, work, (int)nb, C, smooth, dw, out, sums);
  if (dz) {
    long gb = (npix + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(mc_bwd_kernel<C>, dim3((int)gb), dim3(256), 0, s, z, t, npix, HW, ignore, sums, smooth, dw,
                       scale, dz);
  }
}

}  // namespace

// z: float[B][C][HW] logits, target: int64[B][HW] class indices; out: float[3] = {seg, dice, ce};
// dz (optional): float[B][C][HW]; work: double[(SEG_BLOCKS + 1) * 26]; 2 <= C <= 8.  Targets outside [0,C) that
// are not ignore_index are the caller's error (the reference raises in F.one_hot); here they contribute nothing.
extern "C" int s2s_seg_loss_multiclass(const float* z, const long* target, float* dz, float* out, double* work,
                                       long B, long HW, int C, int ignore_index, float smooth, float dice_weight,
                                       float grad_scale, void* stream) {
  if (!z || !target || !out || !work) return S2S_ERR_NULL;
  if (B <= 0 || HW <= 0 || C < 2 || C > MC_MAX) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const long npix = B * HW;
#define S2S_MC(CC) case CC: mc_launch<CC>(z, target, dz, out, work, npix, HW, ignore_index, smooth, dice_weight, grad_scale, s); break;
  switch (C) { S2S_MC(2) S2S_MC(3) S2S_MC(4) S2S_MC(5) S2S_MC(6) S2S_MC(7) S2S_MC(8) }
#undef S2S_MC
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// z, g: float[n] (logits, {0,1} mask); out: float[3] = {seg, dice, bce}; dz (optional): float[n] = scale * d seg / dz;
// work: double[SEG_BLOCKS*4 + 4]
extern "C" int s2s_seg_loss(const float* z, const float* g, float* dz, float* out, double* work, long n, float smooth,
                            float dice_weight, float grad_scale, void* stream) {
  if (!z || !g || !out || !work) return S2S_ERR_NULL;
  if (n <= 0) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  long nb = (n + 255) / 256;
  if (nb > SEG_BLOCKS) nb = SEG_BLOCKS;
  double* sums = work + (long)SEG_BLOCKS * 4;
  hipLaunchKernelGGL(seg_loss_reduce_kernel, dim3((int)nb), dim3(256), 0, s, z, g, n, work);
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(64), 0, s, work, (int)nb, n, smooth, dice_weight, out, sums);
  if (dz) {
    long gb = (n + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3((int)gb), dim3(256), 0, s, z, g, n, sums, smooth, dice_weight,
                       grad_scale, dz);
  }
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
