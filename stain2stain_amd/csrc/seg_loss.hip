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

}  // namespace

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
