// Fused InstanceNorm2d + LeakyReLU (forward and backward), NHWC, HBM-bound.
//
// Row a13 of the scope table: the norm + activation pair of the pix2pix generator /
// PatchGAN discriminator that BASELINE.json's north_star names.  The reference repository holds no such model
// (SURVEY.md F1), so there is no reference file to cite: the semantics are torch's
// nn.InstanceNorm2d(C, eps=1e-5, affine=False|True, track_running_stats=False) followed by
// nn.LeakyReLU(negative_slope) (slope 0 = the decoder's ReLU), and the tests compare against exactly those.
//
//   forward : per-(sample, channel) partial (sum, sum^2) over pixel blocks -> finalize (mean, 1/std, folded
//             scale/shift) -> one pass y = lrelu(x*scale + shift).  Three tensor passes (x twice, y once).
//   backward: dz = g * lrelu'(z) recomputed from x; partial (sum dz, sum dz*xhat) -> finalize (the two
//             per-(sample, channel) means; dgamma/dbeta = sums over the batch when affine) ->
//             dx = gamma*invstd*(dz - c1 - xhat*c2).  Five tensor passes, like the BatchNorm backward of
//             norm_act.hip, with the statistics indexed per sample.
//
// Layout and thread mapping follow norm_act.hip: [B][H][W][C] views with an explicit pixel stride, every
// thread moves 16-B (bf16) / 32-B (fp32) pieces (8 channels of one pixel); partial sums are channel-major
// ([2][B*C][blocks]) so that the finalize reads each (sample, channel) row contiguously.
#include "common.h"
#include <type_traits>

namespace {

__host__ __device__ inline int in_pcb(int C) {       // channel pieces (of 8) per workgroup
  int p = 1;
  while (p < 32 && p * 8 < C) p <<= 1;
  return p;
}

// BWD: g is the gradient wrt y = lrelu(z), g2 (optional) the gradient wrt the second output relu(z) of the forward's
// apply pass: dz = z > 0 ? g + g2 : slope * g.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void in_reduce_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ g, int ldg,
                                                        const T* __restrict__ g2, int ldg2,
                                                        const float* __restrict__ stats, float* __restrict__ part, int B,
                                                        int HW, int C, float slope) {
  const int PCB = in_pcb(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int n = blockIdx.z, c8 = (blockIdx.x * PCB + pc) * 8;
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  if (c8 < C) {
    float mu[8], is[8], sc[8], sh[8];
    if (BWD) {
      const long BC = (long)B * C, o = (long)n * C + c8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { mu[k] = stats[o + k]; is[k] = stats[BC + o + k]; sc[k] = stats[2 * BC + o + k]; sh[k] = stats[3 * BC + o + k]; }
    }
    const T* const xb = x + (long)n * HW * ldx + c8;
    const T* const gb = BWD ? g + (long)n * HW * ldg + c8 : nullptr;
    const T* const gb2 = (BWD && g2) ? g2 + (long)n * HW * ldg2 + c8 : nullptr;
    for (int p = blockIdx.y * WL + wl; p < HW; p += gridDim.y * WL) {
      const f32x8 v = load8(xb + (long)p * ldx);
      if (BWD) {
        const f32x8 gv = load8(gb + (long)p * ldg);
        f32x8 gw;
        if (gb2) gw = load8(gb2 + (long)p * ldg2);
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) gw.v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dz = fmaf(v.v[k], sc[k], sh[k]) > 0.f ? gv.v[k] + gw.v[k] : slope * gv.v[k];
          s1[k] += dz;
          s2[k] = fmaf(dz, (v.v[k] - mu[k]) * is[k], s2[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] += v.v[k]; s2[k] = fmaf(v.v[k], v.v[k], s2[k]); }
      }
    }
  }
  __shared__ float red[2][2304];
  const int rowlen = PCB * 8 + 1;
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][wl * rowlen + pc * 8 + k] = s1[k]; red[1][wl * rowlen + pc * 8 + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * PCB * 8; i += 256) {
    const int which = i / (PCB * 8), cl = i - which * (PCB * 8);
    float s = 0.f;
    for (int r = 0; r < WL; ++r) s += red[which][r * rowlen + cl];
    const int c = blockIdx.x * PCB * 8 + cl;
    if (c < C) part[(((long)which * B + n) * C + c) * gridDim.y + blockIdx.y] = s;
  }
}

// one wave per (sample, channel): row of nblk partials -> the two sums in double
__device__ __forceinline__ void in_row_sums(const float* __restrict__ part, long BC, long i, int nblk, int lane, double& a,
                                            double& b) {
  a = 0.0; b = 0.0;
  for (int k = lane; k < nblk; k += 64) { a += (double)part[i * nblk + k]; b += (double)part[(BC + i) * nblk + k]; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
}

// stats[4][B][C] = mean, invstd, scale (= gamma*invstd), shift (= beta - mean*scale)
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* __restrict__ part, int nblk, int B, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ stats) {
  const long BC = (long)B * C, i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= BC) return;
  double a, b;
  in_row_sums(part, BC, i, nblk, threadIdx.x & 63, a, b);
  if ((threadIdx.x & 63) == 0) {
    const int c = (int)(i % C);
    const double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = (gamma ? gamma[c] : 1.f) * invstd;
    stats[i] = (float)mean;
    stats[BC + i] = invstd;
    stats[2 * BC + i] = sc;
    stats[3 * BC + i] = (beta ? beta[c] : 0.f) - (float)mean * sc;
  }
}

// cs[2][B][C] = per-(sample, channel) sums of dz and dz*xhat
__global__ __launch_bounds__(256) void in_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int B, int C,
                                                              float* __restrict__ cs) {
  const long BC = (long)B * C, i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= BC) return;
  double a, b;
  in_row_sums(part, BC, i, nblk, threadIdx.x & 63, a, b);
  if ((threadIdx.x & 63) == 0) { cs[i] = (float)a; cs[BC + i] = (float)b; }
}

// dbeta[c] (+)= sum_n cs[0][n][c], dgamma[c] (+)= sum_n cs[1][n][c]
__global__ void in_affine_grad_kernel(const float* __restrict__ cs, int B, int C, float* dgamma, float* dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int n = 0; n < B; ++n) { a += (double)cs[(long)n * C + c]; b += (double)cs[((long)B + n) * C + c]; }
  dbeta[c] = accumulate ? dbeta[c] + (float)a : (float)a;
  dgamma[c] = accumulate ? dgamma[c] + (float)b : (float)b;
}

template <typename T>
__global__ __launch_bounds__(256) void in_lrelu_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ stats,
                                                             T* __restrict__ y, int ldy, T* __restrict__ y2, int ldy2, int B,
                                                             int HW, int C, float slope) {
  const int PCB = in_pcb(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int n = blockIdx.z, c8 = (blockIdx.x * PCB + pc) * 8;
  if (c8 >= C) return;
  const long BC = (long)B * C, o = (long)n * C + c8;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = stats[2 * BC + o + k]; sh[k] = stats[3 * BC + o + k]; }
  const T* const xb = x + (long)n * HW * ldx + c8;
  T* const yb = y + (long)n * HW * ldy + c8;
  T* const yb2 = y2 ? y2 + (long)n * HW * ldy2 + c8 : nullptr;      // relu(z): the skip tensor in the decoder's cat buffer
  for (int p = blockIdx.y * WL + wl; p < HW; p += gridDim.y * WL) {
    const f32x8 v = load8(xb + (long)p * ldx);
    f32x8 o8, r8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = fmaf(v.v[k], sc[k], sh[k]);
      o8.v[k] = z > 0.f ? z : slope * z;
      r8.v[k] = fmaxf(z, 0.f);
    }
    store8(yb + (long)p * ldy, o8);
    if (yb2) store8(yb2 + (long)p * ldy2, r8);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void in_lrelu_bwd_apply_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ g2, int ldg2,
                                                                 const T* __restrict__ x, int ldx,
                                                                 const float* __restrict__ stats, const float* __restrict__ cs,
                                                                 T* __restrict__ dx, int lddx, int B, int HW, int C,
                                                                 float slope) {
  const int PCB = in_pcb(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int n = blockIdx.z, c8 = (blockIdx.x * PCB + pc) * 8;
  if (c8 >= C) return;
  const long BC = (long)B * C, o = (long)n * C + c8;
  const float inv_hw = 1.f / (float)HW;
  float mu[8], is[8], sc[8], sh[8], k1[8], k2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = stats[o + k]; is[k] = stats[BC + o + k]; sc[k] = stats[2 * BC + o + k]; sh[k] = stats[3 * BC + o + k];
    k1[k] = cs[o + k] * inv_hw; k2[k] = cs[BC + o + k] * inv_hw;
  }
  const T* const xb = x + (long)n * HW * ldx + c8;
  const T* const gb = g + (long)n * HW * ldg + c8;
  const T* const gb2 = g2 ? g2 + (long)n * HW * ldg2 + c8 : nullptr;
  T* const db = dx + (long)n * HW * lddx + c8;
  for (int p = blockIdx.y * WL + wl; p < HW; p += gridDim.y * WL) {
    const f32x8 v = load8(xb + (long)p * ldx);
    const f32x8 gv = load8(gb + (long)p * ldg);
    f32x8 gw;
    if (gb2) gw = load8(gb2 + (long)p * ldg2);
    else {
#pragma unroll
      for (int k = 0; k < 8; ++k) gw.v[k] = 0.f;
    }
    f32x8 o8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dz = fmaf(v.v[k], sc[k], sh[k]) > 0.f ? gv.v[k] + gw.v[k] : slope * gv.v[k];
      o8.v[k] = sc[k] * (dz - k1[k] - ((v.v[k] - mu[k]) * is[k]) * k2[k]);      // sc = gamma * invstd
    }
    store8(db + (long)p * lddx, o8);
  }
}

// ---- single-launch forms for small maps ------------------------------------------------------------------------------
// Up to 32x32 pixels a (sample, 32-channel) slice is at most 64 KiB (bf16): one workgroup reads it twice (the second
// time from L1 / L2), with the reduction in between done in LDS -- one launch instead of reduce + finalize + apply.  At
// these sizes the three-launch form is bound by launch latency, not bandwidth (5-14 us per launch for < 10 MB), and the
// inner U-Net levels and the PatchGAN's last layers run 14 such norms forward and backward per step.
constexpr int IN_SMALL_HW = 1024;
constexpr int IN_CG = 32;                      // channels per workgroup: 4 pieces x 64 pixel lanes

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void in_small_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ g, int ldg,
                                                       const T* __restrict__ g2, int ldg2, T* __restrict__ y, int ldy,
                                                       T* __restrict__ y2, int ldy2, float* __restrict__ stats, int B, int HW,
                                                       int C, float eps, float slope) {
  constexpr int PCB = IN_CG / 8, WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int n = blockIdx.y, c8 = blockIdx.x * IN_CG + pc * 8;
  const bool live = c8 < C;
  const long BC = (long)B * C, o = (long)n * C + c8;
  const T* const xb = x + (long)n * HW * ldx + c8;
  __shared__ double red[2][WL][IN_CG + 1];
  __shared__ float fin[4][IN_CG];               // forward: mean, invstd, scale, shift; backward: k1, k2 in rows 0, 1
  float mu[8], is[8], sc[8], sh[8];
  if (BWD && live) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { mu[k] = stats[o + k]; is[k] = stats[BC + o + k]; sc[k] = stats[2 * BC + o + k]; sh[k] = stats[3 * BC + o + k]; }
  }
  float s1[8], s2[8], shift0[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; shift0[k] = 0.f; }
  if (live && !BWD) {
    // shifted sums: sum(x - K), sum((x - K)^2) with K = the channel's first pixel.  E[x^2] - mean^2 loses everything
    // when the spread of a small map is far below its level (a 2-pixel map with var ~ eps); shifting by a value of the
    // data removes that cancellation at no cost
    const f32x8 v0 = load8(xb);
#pragma unroll
    for (int k = 0; k < 8; ++k) shift0[k] = v0.v[k];
  }
  if (live) {
    for (int p = wl; p < HW; p += WL) {
      f32x8 v = load8(xb + (long)p * ldx);
      if (!BWD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v.v[k] -= shift0[k];
      }
      if (BWD) {
        const f32x8 gv = load8(g + ((long)n * HW + p) * ldg + c8);
        f32x8 gw;
        if (g2) gw = load8(g2 + ((long)n * HW + p) * ldg2 + c8);
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) gw.v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dz = fmaf(v.v[k], sc[k], sh[k]) > 0.f ? gv.v[k] + gw.v[k] : slope * gv.v[k];
          s1[k] += dz;
          s2[k] = fmaf(dz, (v.v[k] - mu[k]) * is[k], s2[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] += v.v[k]; s2[k] = fmaf(v.v[k], v.v[k], s2[k]); }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][wl][pc * 8 + k] = (double)s1[k]; red[1][wl][pc * 8 + k] = (double)s2[k]; }
  if (!BWD && wl == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) fin[0][pc * 8 + k] = shift0[k];           // K per channel, read back by the finalising lane
  }
  __syncthreads();
  if (threadIdx.x < IN_CG) {
    double a = 0.0, b = 0.0;
    for (int r = 0; r < WL; ++r) { a += red[0][r][threadIdx.x]; b += red[1][r][threadIdx.x]; }
    const int c = blockIdx.x * IN_CG + threadIdx.x;
    if (BWD) {
      fin[0][threadIdx.x] = (float)a / (float)HW;            // as the three-launch form: float sums, then * 1/HW
      fin[1][threadIdx.x] = (float)b / (float)HW;
    } else {
      const double ms = a / (double)HW;                       // mean of the shifted values
      double var = b / (double)HW - ms * ms;
      if (var < 0.0) var = 0.0;
      const double mean = ms + (double)fin[0][threadIdx.x];
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      fin[0][threadIdx.x] = (float)mean; fin[1][threadIdx.x] = invstd;
      fin[2][threadIdx.x] = invstd; fin[3][threadIdx.x] = -(float)mean * invstd;
      if (c < C) {
        const long i = (long)n * C + c;
        stats[i] = (float)mean; stats[BC + i] = invstd; stats[2 * BC + i] = invstd; stats[3 * BC + i] = -(float)mean * invstd;
      }
    }
  }
  __syncthreads();
  if (!live) return;
  if (BWD) {
    float k1[8], k2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { k1[k] = fin[0][pc * 8 + k]; k2[k] = fin[1][pc * 8 + k]; }
    for (int p = wl; p < HW; p += WL) {
      const f32x8 v = load8(xb + (long)p * ldx);
      const f32x8 gv = load8(g + ((long)n * HW + p) * ldg + c8);
      f32x8 gw;
      if (g2) gw = load8(g2 + ((long)n * HW + p) * ldg2 + c8);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) gw.v[k] = 0.f;
      }
      f32x8 o8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dz = fmaf(v.v[k], sc[k], sh[k]) > 0.f ? gv.v[k] + gw.v[k] : slope * gv.v[k];
        o8.v[k] = sc[k] * (dz - k1[k] - ((v.v[k] - mu[k]) * is[k]) * k2[k]);
      }
      store8(y + ((long)n * HW + p) * ldy + c8, o8);
    }
  } else {
    float fs[8], fh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { fs[k] = fin[2][pc * 8 + k]; fh[k] = fin[3][pc * 8 + k]; }
    for (int p = wl; p < HW; p += WL) {
      const f32x8 v = load8(xb + (long)p * ldx);
      f32x8 o8, r8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = fmaf(v.v[k], fs[k], fh[k]);
        o8.v[k] = z > 0.f ? z : slope * z;
        r8.v[k] = fmaxf(z, 0.f);
      }
      store8(y + ((long)n * HW + p) * ldy + c8, o8);
      if (y2) store8(y2 + ((long)n * HW + p) * ldy2 + c8, r8);
    }
  }
}

// bf16 form of the above with the slice held in registers: a (sample, 64-channel) slice of <= 1024 pixels is 16 pieces
// of 16 B per thread (512 threads: 8 channel pieces x 64 pixel lanes), so every load of the launch is issued up front
// (full memory-level parallelism, whole 128-B lines per pixel) and the second pass needs no memory at all.  x, g and g2
// stay in their bf16 form (4 VGPRs per piece); the arithmetic is the one of in_small_kernel.
constexpr int IN_RCG = 64;
constexpr int IN_RIT = IN_SMALL_HW / 64;

__device__ __forceinline__ f32x8 cvt8(const bf16x8& a) {
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
  return r;
}

// SPLIT: the tensor a split-K convolution would have produced is folded here from its fp32 partial slabs
// slabs[S][B*HW][C] (+ bias) instead of by a reduce launch of its own -- the forward's input x (the folded, bf16-rounded
// tensor is also written to xw: the backward reads it), the backward's incoming gradient g.  Same order of additions and
// the same rounding as convk_splitk_reduce_kernel, so the results are those of the two-launch form bit for bit.
struct InSplit {
  const float* slabs;
  const float* bias;
  long stride;            // elements between slabs
  int S;
  bf16_t* xw;             // forward: where the folded tensor goes (pixel stride ldx)
};

__device__ __forceinline__ bf16x8 in_fold_piece(const InSplit& sp, long elem, int c8) {
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = sp.bias ? sp.bias[c8 + k] : 0.f;
  for (int z = 0; z < sp.S; ++z) {
    const f32x8 v = load8(sp.slabs + (long)z * sp.stride + elem);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v.v[k];
  }
  bf16x8 r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = (bf16_t)acc[k];
  return r;
}

template <bool BWD, bool G2, bool SPLIT = false>
__global__ __launch_bounds__(512) void in_small_res_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ g,
                                                           int ldg, const bf16_t* __restrict__ g2, int ldg2,
                                                           bf16_t* __restrict__ y, int ldy, bf16_t* __restrict__ y2, int ldy2,
                                                           float* __restrict__ stats, int B, int HW, int C, float eps,
                                                           float slope, InSplit sp = InSplit{}) {
  const int pc = threadIdx.x & 7, wl = threadIdx.x >> 3, wave = threadIdx.x >> 6;
  const int n = blockIdx.y, c8 = blockIdx.x * IN_RCG + pc * 8;
  const bool live = c8 < C;
  const long BC = (long)B * C, o = (long)n * C + c8;
  const bf16_t* const xb = x + (long)n * HW * ldx + c8;
  __shared__ double red[2][8][IN_RCG];
  __shared__ float fin[4][IN_RCG];
  bf16x8 xr[IN_RIT], gr[BWD ? IN_RIT : 1];     // g2 is re-read in both passes (x, g and g2 resident would spill)
  float mu[8], is[8], sc[8], sh[8], shift0[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) shift0[k] = 0.f;
  if (live) {
#pragma unroll
    for (int it = 0; it < IN_RIT; ++it) {
      const int p = wl + it * 64;
      if (p < HW) {
        if (SPLIT && !BWD) {
          xr[it] = in_fold_piece(sp, ((long)n * HW + p) * C + c8, c8);
          *reinterpret_cast<bf16x8*>(sp.xw + ((long)n * HW + p) * ldx + c8) = xr[it];
        } else {
          xr[it] = *reinterpret_cast<const bf16x8*>(xb + (long)p * ldx);
        }
        if (BWD) {
          if (SPLIT) gr[it] = in_fold_piece(sp, ((long)n * HW + p) * C + c8, c8);
          else gr[it] = *reinterpret_cast<const bf16x8*>(g + ((long)n * HW + p) * ldg + c8);
        }
      }
    }
    if (BWD) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { mu[k] = stats[o + k]; is[k] = stats[BC + o + k]; sc[k] = stats[2 * BC + o + k]; sh[k] = stats[3 * BC + o + k]; }
    } else {
      // the shift of in_small_kernel: the channel's first pixel
      const f32x8 v0 = SPLIT ? cvt8(in_fold_piece(sp, (long)n * HW * C + c8, c8)) : load8(xb);
#pragma unroll
      for (int k = 0; k < 8; ++k) shift0[k] = v0.v[k];
    }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  if (live) {
#pragma unroll
    for (int it = 0; it < IN_RIT; ++it) {
      if (wl + it * 64 < HW) {
        const f32x8 v = cvt8(xr[it]);
        if (BWD) {
          const f32x8 gv = cvt8(gr[it]);
          bf16x8 hr;
          if (G2) hr = *reinterpret_cast<const bf16x8*>(g2 + ((long)n * HW + wl + it * 64) * ldg2 + c8);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float gw = G2 ? (float)hr[k] : 0.f;
            const float dz = fmaf(v.v[k], sc[k], sh[k]) > 0.f ? gv.v[k] + gw : slope * gv.v[k];
            s1[k] += dz;
            s2[k] = fmaf(dz, (v.v[k] - mu[k]) * is[k], s2[k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) { const float d = v.v[k] - shift0[k]; s1[k] += d; s2[k] = fmaf(d, d, s2[k]); }
        }
      }
    }
  }
  // the 8 pixel lanes of a wave that share a channel piece, then the 8 waves through LDS
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double a = (double)s1[k], b = (double)s2[k];
#pragma unroll
    for (int m = 8; m <= 32; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
    if ((threadIdx.x & 63) < 8) { red[0][wave][pc * 8 + k] = a; red[1][wave][pc * 8 + k] = b; }
  }
  if (!BWD && wl == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) fin[0][pc * 8 + k] = shift0[k];
  }
  __syncthreads();
  if (threadIdx.x < IN_RCG) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a += red[0][r][threadIdx.x]; b += red[1][r][threadIdx.x]; }
    const int c = blockIdx.x * IN_RCG + threadIdx.x;
    if (BWD) {
      fin[0][threadIdx.x] = (float)a / (float)HW;
      fin[1][threadIdx.x] = (float)b / (float)HW;
    } else {
      const double ms = a / (double)HW;
      double var = b / (double)HW - ms * ms;
      if (var < 0.0) var = 0.0;
      const double mean = ms + (double)fin[0][threadIdx.x];
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      fin[0][threadIdx.x] = (float)mean; fin[1][threadIdx.x] = invstd;
      fin[2][threadIdx.x] = invstd; fin[3][threadIdx.x] = -(float)mean * invstd;
      if (c < C) {
        const long i = (long)n * C + c;
        stats[i] = (float)mean; stats[BC + i] = invstd; stats[2 * BC + i] = invstd; stats[3 * BC + i] = -(float)mean * invstd;
      }
    }
  }
  __syncthreads();
  if (!live) return;
  if (BWD) {
    float k1[8], k2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { k1[k] = fin[0][pc * 8 + k]; k2[k] = fin[1][pc * 8 + k]; }
#pragma unroll
    for (int it = 0; it < IN_RIT; ++it) {
      const int p = wl + it * 64;
      if (p < HW) {
        const f32x8 v = cvt8(xr[it]), gv = cvt8(gr[it]);
        bf16x8 hr;
        if (G2) hr = *reinterpret_cast<const bf16x8*>(g2 + ((long)n * HW + p) * ldg2 + c8);
        f32x8 o8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gw = G2 ? (float)hr[k] : 0.f;
          const float dz = fmaf(v.v[k], sc[k], sh[k]) > 0.f ? gv.v[k] + gw : slope * gv.v[k];
          o8.v[k] = sc[k] * (dz - k1[k] - ((v.v[k] - mu[k]) * is[k]) * k2[k]);
        }
        store8(y + ((long)n * HW + p) * ldy + c8, o8);
      }
    }
  } else {
    float fs[8], fh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { fs[k] = fin[2][pc * 8 + k]; fh[k] = fin[3][pc * 8 + k]; }
#pragma unroll
    for (int it = 0; it < IN_RIT; ++it) {
      const int p = wl + it * 64;
      if (p < HW) {
        const f32x8 v = cvt8(xr[it]);
        f32x8 o8, r8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float z = fmaf(v.v[k], fs[k], fh[k]);
          o8.v[k] = z > 0.f ? z : slope * z;
          r8.v[k] = fmaxf(z, 0.f);
        }
        store8(y + ((long)n * HW + p) * ldy + c8, o8);
        if (y2) store8(y2 + ((long)n * HW + p) * ldy2 + c8, r8);
      }
    }
  }
}

inline int in_blocks(int B, int H, int W, int C) {
  const int pcb = in_pcb(C), wl = 256 / pcb, groups = cdiv(C / 8, pcb);
  const long hw = (long)H * W;
  long nb = cdiv(4096, groups * B);                 // ~16 workgroups per CU over the whole launch
  const long most = (hw + wl - 1) / wl;
  if (nb > most) nb = most;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}

inline bool in_args_ok(int B, int H, int W, int C, int ld0, int ld1) {
  return B > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0 && (ld0 % 8) == 0 && (ld1 % 8) == 0 && B <= 65535 &&
         (long)H * W < (1L << 31) && (long)B * C < (1L << 31);
}

}  // namespace

extern "C" int s2s_instnorm_blocks(int B, int H, int W, int C) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8)) return S2S_ERR_SHAPE;
  return in_blocks(B, H, W, C);
}

// y2 (optional, pixel stride ldy2): a second output relu(z) -- the pix2pix generator's ReLU'd skip tensor, written
// straight into the decoder's concatenation buffer by the same pass.
extern "C" int s2s_instnorm_lrelu_fwd2(int dtype, const void* x, int ldx, const float* gamma, const float* beta, void* y,
                                       int ldy, void* y2, int ldy2, float* work, float* stats, int B, int H, int W, int C,
                                       float eps, float slope, void* stream) {
  if (!x || !y || !work || !stats) return S2S_ERR_NULL;
  if ((gamma == nullptr) != (beta == nullptr)) return S2S_ERR_NULL;
  if (!in_args_ok(B, H, W, C, ldx, ldy) || (y2 && (ldy2 % 8))) return S2S_ERR_SHAPE;
  if ((long)H * W <= 1) return S2S_ERR_SHAPE;      // torch raises for a single spatial element in training mode
  hipStream_t s = (hipStream_t)stream;
  const int nb = in_blocks(B, H, W, C), HW = H * W;
  if (!gamma && HW <= IN_SMALL_HW) {            // small maps: one launch (in_small_kernel)
    const dim3 sg(cdiv(C, IN_CG), B), rg(cdiv(C, IN_RCG), B);
    if (dtype == S2S_BF16)
      hipLaunchKernelGGL((in_small_res_kernel<false, false>), rg, dim3(512), 0, s, (const bf16_t*)x, ldx, (const bf16_t*)nullptr,
                         0, (const bf16_t*)nullptr, 0, (bf16_t*)y, ldy, (bf16_t*)y2, ldy2, stats, B, HW, C, eps, slope);
    else if (dtype == S2S_F32)
      hipLaunchKernelGGL((in_small_kernel<float, false>), sg, dim3(256), 0, s, (const float*)x, ldx, (const float*)nullptr, 0,
                         (const float*)nullptr, 0, (float*)y, ldy, (float*)y2, ldy2, stats, B, HW, C, eps, slope);
    else return S2S_ERR_DTYPE;
    S2S_LAUNCH_CHECK();
    return S2S_OK;
  }
  const dim3 grid(cdiv(C / 8, in_pcb(C)), nb, B);
  const unsigned fin = (unsigned)(((long)B * C + 3) / 4);
#define S2S_IN_FWD(TT)                                                                                              \
  hipLaunchKernelGGL((in_reduce_kernel<TT, false>), grid, dim3(256), 0, s, (const TT*)x, ldx, (const TT*)nullptr, 0, \
                     (const TT*)nullptr, 0, (const float*)nullptr, work, B, HW, C, 0.f);                             \
  hipLaunchKernelGGL(in_finalize_kernel, dim3(fin), dim3(256), 0, s, work, nb, B, C, (double)HW, gamma, beta, eps,   \
                     stats);                                                                                         \
  hipLaunchKernelGGL(in_lrelu_apply_kernel<TT>, grid, dim3(256), 0, s, (const TT*)x, ldx, stats, (TT*)y, ldy,        \
                     (TT*)y2, ldy2, B, HW, C, slope)
  if (dtype == S2S_BF16) { S2S_IN_FWD(bf16_t); }
  else if (dtype == S2S_F32) { S2S_IN_FWD(float); }
  else return S2S_ERR_DTYPE;
#undef S2S_IN_FWD
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// The single-launch forms with the input folded from a split-K convolution's partial slabs (bf16, no affine, maps of at
// most 32 x 32: s2s_instnorm_split_ok).  slabs: float[S][B*H*W][C] as s2s_conv4x4s2_nhwc / s2s_convt4x4s2_nhwc leave them
// with y == NULL; bias: the convolution's (or NULL).
extern "C" int s2s_instnorm_split_ok(int dtype, int H, int W, int C) {
  return dtype == S2S_BF16 && H > 0 && W > 0 && (long)H * W <= IN_SMALL_HW && (long)H * W > 1 && C > 0 && (C % 8) == 0;
}

// forward: raw (bf16, pixel stride ldraw) <- fold(slabs) + bias; then as s2s_instnorm_lrelu_fwd2 on raw
extern "C" int s2s_instnorm_lrelu_fwd_split(int dtype, const float* slabs, int S, const float* bias, void* raw, int ldraw,
                                            void* y, int ldy, void* y2, int ldy2, float* stats, int B, int H, int W, int C,
                                            float eps, float slope, void* stream) {
  if (!slabs || !raw || !y || !stats) return S2S_ERR_NULL;
  if (!s2s_instnorm_split_ok(dtype, H, W, C) || S < 1 || !in_args_ok(B, H, W, C, ldraw, ldy) || (y2 && (ldy2 % 8)))
    return S2S_ERR_SHAPE;
  if (((uintptr_t)slabs & 15) || ((uintptr_t)raw & 15) || ((uintptr_t)y & 15) || ((uintptr_t)y2 & 15)) return S2S_ERR_ALIGN;
  const int HW = H * W;
  InSplit sp{slabs, bias, (long)B * HW * C, S, (bf16_t*)raw};
  hipLaunchKernelGGL((in_small_res_kernel<false, false, true>), dim3(cdiv(C, IN_RCG), B), dim3(512), 0, (hipStream_t)stream,
                     (const bf16_t*)raw, ldraw, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, (bf16_t*)y, ldy,
                     (bf16_t*)y2, ldy2, stats, B, HW, C, eps, slope, sp);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// backward: g <- fold(gslabs) (bf16-rounded, no bias); then as s2s_instnorm_lrelu_bwd2
extern "C" int s2s_instnorm_lrelu_bwd_split(int dtype, const float* gslabs, int S, const void* g2, int ldg2, const void* x,
                                            int ldx, const float* stats, void* dx, int lddx, int B, int H, int W, int C,
                                            float slope, void* stream) {
  if (!gslabs || !x || !stats || !dx) return S2S_ERR_NULL;
  if (!s2s_instnorm_split_ok(dtype, H, W, C) || S < 1 || !in_args_ok(B, H, W, C, ldx, lddx) || (g2 && (ldg2 % 8)))
    return S2S_ERR_SHAPE;
  if (((uintptr_t)gslabs & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)g2 & 15)) return S2S_ERR_ALIGN;
  const int HW = H * W;
  InSplit sp{gslabs, nullptr, (long)B * HW * C, S, nullptr};
  const dim3 rg(cdiv(C, IN_RCG), B);
  if (g2)
    hipLaunchKernelGGL((in_small_res_kernel<true, true, true>), rg, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)nullptr, 0, (const bf16_t*)g2, ldg2, (bf16_t*)dx, lddx, (bf16_t*)nullptr, 0,
                       const_cast<float*>(stats), B, HW, C, 0.f, slope, sp);
  else
    hipLaunchKernelGGL((in_small_res_kernel<true, false, true>), rg, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, (bf16_t*)dx, lddx, (bf16_t*)nullptr, 0,
                       const_cast<float*>(stats), B, HW, C, 0.f, slope, sp);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_instnorm_lrelu_fwd(int dtype, const void* x, int ldx, const float* gamma, const float* beta, void* y,
                                      int ldy, float* work, float* stats, int B, int H, int W, int C, float eps,
                                      float slope, void* stream) {
  return s2s_instnorm_lrelu_fwd2(dtype, x, ldx, gamma, beta, y, ldy, nullptr, 8, work, stats, B, H, W, C, eps, slope,
                                 stream);
}

// g2 (optional): the gradient wrt the forward's second output relu(z); dz = z > 0 ? g + g2 : slope * g.
extern "C" int s2s_instnorm_lrelu_bwd2(int dtype, const void* g, int ldg, const void* g2, int ldg2, const void* x, int ldx,
                                       const float* stats, void* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                                       float* work, int B, int H, int W, int C, float slope, void* stream) {
  if (!g || !x || !stats || !dx || !work) return S2S_ERR_NULL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return S2S_ERR_NULL;
  if (!in_args_ok(B, H, W, C, ldx, ldg) || (lddx % 8) || (g2 && (ldg2 % 8))) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int nb = in_blocks(B, H, W, C), HW = H * W;
  if (!dgamma && HW <= IN_SMALL_HW) {           // small maps: one launch (in_small_kernel)
    const dim3 sg(cdiv(C, IN_CG), B), rg(cdiv(C, IN_RCG), B);
    if (dtype == S2S_BF16 && g2)
      hipLaunchKernelGGL((in_small_res_kernel<true, true>), rg, dim3(512), 0, s, (const bf16_t*)x, ldx, (const bf16_t*)g, ldg,
                         (const bf16_t*)g2, ldg2, (bf16_t*)dx, lddx, (bf16_t*)nullptr, 0, const_cast<float*>(stats), B, HW, C,
                         0.f, slope);
    else if (dtype == S2S_BF16)
      hipLaunchKernelGGL((in_small_res_kernel<true, false>), rg, dim3(512), 0, s, (const bf16_t*)x, ldx, (const bf16_t*)g, ldg,
                         (const bf16_t*)nullptr, 0, (bf16_t*)dx, lddx, (bf16_t*)nullptr, 0, const_cast<float*>(stats), B, HW, C,
                         0.f, slope);
    else if (dtype == S2S_F32)
      hipLaunchKernelGGL((in_small_kernel<float, true>), sg, dim3(256), 0, s, (const float*)x, ldx, (const float*)g, ldg,
                         (const float*)g2, ldg2, (float*)dx, lddx, (float*)nullptr, 0, const_cast<float*>(stats), B, HW, C,
                         0.f, slope);
    else return S2S_ERR_DTYPE;
    S2S_LAUNCH_CHECK();
    return S2S_OK;
  }
  float* const part = work;                                   // [2][B*C][nb]
  float* const cs = work + (size_t)2 * B * C * nb;            // [2][B][C]
  const dim3 grid(cdiv(C / 8, in_pcb(C)), nb, B);
  const unsigned fin = (unsigned)(((long)B * C + 3) / 4);
#define S2S_IN_BWD(TT)                                                                                              \
  hipLaunchKernelGGL((in_reduce_kernel<TT, true>), grid, dim3(256), 0, s, (const TT*)x, ldx, (const TT*)g, ldg,      \
                     (const TT*)g2, ldg2, stats, part, B, HW, C, slope);                                             \
  hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3(fin), dim3(256), 0, s, part, nb, B, C, cs);                        \
  if (dgamma)                                                                                                        \
    hipLaunchKernelGGL(in_affine_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, cs, B, C, dgamma, dbeta,          \
                       accumulate);                                                                                  \
  hipLaunchKernelGGL(in_lrelu_bwd_apply_kernel<TT>, grid, dim3(256), 0, s, (const TT*)g, ldg, (const TT*)g2, ldg2,   \
                     (const TT*)x, ldx, stats, cs, (TT*)dx, lddx, B, HW, C, slope)
  if (dtype == S2S_BF16) { S2S_IN_BWD(bf16_t); }
  else if (dtype == S2S_F32) { S2S_IN_BWD(float); }
  else return S2S_ERR_DTYPE;
#undef S2S_IN_BWD
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_instnorm_lrelu_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const float* stats,
                                      void* dx, int lddx, float* dgamma, float* dbeta, int accumulate, float* work,
                                      int B, int H, int W, int C, float slope, void* stream) {
  return s2s_instnorm_lrelu_bwd2(dtype, g, ldg, nullptr, 8, x, ldx, stats, dx, lddx, dgamma, dbeta, accumulate, work, B, H,
                                 W, C, slope, stream);
}

// ---- space-to-depth of the zero-padded image and its inverse (row a13) -------------------------------------------
// The 4x4 stride-2 convolution runs as a 2x2 convolution over xs[n][p][q][(r*2+s)*C + c] = xpad[n][2p+r][2q+s][c],
// xpad = x with a one-pixel zero border (H, W even; xs is (H/2+1) x (W/2+1) x 4C).  One 16-byte piece per thread,
// both directions; the inverse drops the border (the data gradient's border cells hold the padding's gradient).
namespace {

template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void s2d_pad1_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int B,
                                                       int H, int W, int C) {
  const int cp = C >> 3, Hs = H / 2 + 1, Ws = W / 2 + 1;
  const long total = (long)B * H * W * cp;          // one piece of the un-padded image per thread, either direction
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int pc = (int)(i % cp);
    long t = i / cp;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const int u = y + 1, v = x + 1;                  // position in the padded image
    const long cell = ((long)n * Hs + (u >> 1)) * Ws + (v >> 1);
    const int sub = ((u & 1) * 2 + (v & 1)) * C + pc * 8;
    const long pix = ((long)n * H + y) * W + x;
    if (INVERSE) store8(dst + pix * ldd + pc * 8, load8(src + cell * lds_ + sub));     // bf16 <-> fp32 round trip is exact
    else store8(dst + cell * ldd + sub, load8(src + pix * lds_ + pc * 8));
  }
}

// the border sub-cells of xs that no image pixel maps to: row 0 (r = 0), row Hs-1 (r = 1), column 0 (s = 0), column Ws-1 (s = 1)
template <typename T>
__global__ __launch_bounds__(256) void s2d_border_zero_kernel(T* __restrict__ dst, int ldd, int B, int H, int W, int C) {
  const int cp = C >> 3, Hs = H / 2 + 1, Ws = W / 2 + 1;
  const int per_img = 2 * Ws + 2 * Hs;               // border cells visited per image (corners twice: harmless)
  const long total = (long)B * per_img * 2 * cp;     // x the two sub-positions along the border, x pieces
  f32x8 z;
#pragma unroll
  for (int k = 0; k < 8; ++k) z.v[k] = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int pc = (int)(i % cp);
    long t = i / cp;
    const int o = (int)(t % 2); t /= 2;              // the free sub-index along the border
    const int b = (int)(t % per_img);
    const int n = (int)(t / per_img);
    int p, q, r, s;
    if (b < Ws) { p = 0; q = b; r = 0; s = o; }
    else if (b < 2 * Ws) { p = Hs - 1; q = b - Ws; r = 1; s = o; }
    else if (b < 2 * Ws + Hs) { p = b - 2 * Ws; q = 0; r = o; s = 0; }
    else { p = b - 2 * Ws - Hs; q = Ws - 1; r = o; s = 1; }
    store8(dst + (((long)n * Hs + p) * Ws + q) * ldd + (r * 2 + s) * C + pc * 8, z);
  }
}

}  // namespace

extern "C" int s2s_space_to_depth_pad1(int dtype, const void* x, int ldx, void* xs, int ldxs, int inverse, int B, int H,
                                       int W, int C, void* stream) {
  if (!x || !xs) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (H % 2) || (W % 2) || (C % 8) || (ldx % 8) || (ldxs % 8)) return S2S_ERR_SHAPE;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * H * W * (C / 8);
  long nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  const long bt = (long)B * (2 * (W / 2 + 1) + 2 * (H / 2 + 1)) * 2 * (C / 8);
#define S2S_S2D(TT)                                                                                                  \
  if (inverse) { /* x = image-shaped destination, xs = source */                                                      \
    hipLaunchKernelGGL((s2d_pad1_kernel<TT, true>), dim3((unsigned)nb), dim3(256), 0, s, (const TT*)xs, ldxs,        \
                       (TT*)const_cast<void*>(x), ldx, B, H, W, C);                                                   \
  } else {                                                                                                            \
    hipLaunchKernelGGL(s2d_border_zero_kernel<TT>, dim3((unsigned)((bt + 255) / 256)), dim3(256), 0, s, (TT*)xs,      \
                       ldxs, B, H, W, C);                                                                             \
    hipLaunchKernelGGL((s2d_pad1_kernel<TT, false>), dim3((unsigned)nb), dim3(256), 0, s, (const TT*)x, ldx,          \
                       (TT*)xs, ldxs, B, H, W, C);                                                                    \
  }
  if (dtype == S2S_BF16) { S2S_S2D(bf16_t) } else { S2S_S2D(float) }
#undef S2S_S2D
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_pack_conv4x4_t(int dtype, const float* w_oihw, void* wf, void* wd, int Cout, int Cin, int stride,
                                  void* stream);

// ---- weight packing for the 4x4 layers (row a13) -----------------------------------------------------------------
// fp32 master w[Cout][Cin][4][4] -> the two bf16 MFMA operands of s2s_conv2x2_nhwc (stride 2: taps (a,b), k =
// (r*2+s)*Cin + c, w[o][c][2a+r][2b+s]) or s2s_conv4x4s1_nhwc (stride 1: taps (kh,kw), k = c):
//   wf[ceil(K/32)][taps][Cout][32]            forward operand
//   wd[ceil(Cout/32)][taps][K][32]            data-gradient operand: taps flipped, k and o exchanged
// One element per thread; the 64-byte (o, c) blocks of the master are re-read from L2 by the threads that need
// their other taps.
namespace {

__device__ __forceinline__ float w4x4_at(const float* __restrict__ w, int Cin, int o, int k, int tap, int stride2) {
  int c, kh, kw;
  if (stride2) { const int rs = k / Cin; c = k - rs * Cin; kh = 2 * (tap >> 1) + (rs >> 1); kw = 2 * (tap & 1) + (rs & 1); }
  else { c = k; kh = tap >> 2; kw = tap & 3; }
  return w[(((long)o * Cin + c) * 4 + kh) * 4 + kw];
}

// one element of the packed pair: e < nf -> forward operand, else data-gradient operand
template <typename T>
__device__ __forceinline__ void pack4x4_elem(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout, int Cin,
                                             int stride2, long e) {
  const int taps = stride2 ? 4 : 16, K = stride2 ? 4 * Cin : Cin;
  const long nf = (long)((K + 31) / 32) * taps * Cout * 32;
  if (e < nf) {
    const int kk = (int)(e & 31);
    long t = e >> 5;
    const int o = (int)(t % Cout); t /= Cout;
    const int tap = (int)(t % taps);
    const int k = (int)(t / taps) * 32 + kk;
    wf[e] = from_f32<T>(k < K ? w4x4_at(w, Cin, o, k, tap, stride2) : 0.f);
  } else {
    const long d = e - nf;
    const int oo = (int)(d & 31);
    long t = d >> 5;
    const int k = (int)(t % K); t /= K;
    const int tapf = (int)(t % taps);
    const int o = (int)(t / taps) * 32 + oo;
    wd[d] = from_f32<T>(o < Cout ? w4x4_at(w, Cin, o, k, taps - 1 - tapf, stride2) : 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack4x4_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                                      int Cout, int Cin, int stride2) {
  const int taps = stride2 ? 4 : 16, K = stride2 ? 4 * Cin : Cin;
  const long n = (long)((K + 31) / 32) * taps * Cout * 32 + (long)((Cout + 31) / 32) * taps * K * 32;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
    pack4x4_elem<T>(w, wf, wd, Cout, Cin, stride2, e);
}

// every 4x4 layer of a network in one launch (after the fused Adam step): desc[l] = {w, wf, wd, Cout, Cin, stride2,
// first block}.  Layers whose channel counts are multiples of 32 go through an LDS tile: a workgroup owns a 32 (o) x
// 32 (c) block of the master (each o row is 2 KiB contiguous), holds it as tile[tap][o][c] and writes both operands
// as 16-byte (bf16) pieces -- the element-wise form reads 4 bytes out of every 64-byte line it touches.  The few
// 8-channel layers (image side) keep the element-wise form, 2048 elements per workgroup.
__host__ __device__ inline long pack4x4_layer_blocks(int Cout, int Cin, int stride2) {
  if ((Cout % 32) == 0 && (Cin % 32) == 0) return (long)(Cout / 32) * (Cin / 32);
  const int taps = stride2 ? 4 : 16, K = stride2 ? 4 * Cin : Cin;
  const long n = (long)((K + 31) / 32) * taps * Cout * 32 + (long)((Cout + 31) / 32) * taps * K * 32;
  return (n + 2047) / 2048;
}

template <typename T>
__global__ __launch_bounds__(256) void pack4x4_batched_kernel(const long* __restrict__ desc, int nlayers) {
  int l = 0;
  while (l + 1 < nlayers && (long)blockIdx.x >= desc[(l + 1) * 7 + 6]) ++l;
  const long* d = desc + l * 7;
  const float* __restrict__ w = reinterpret_cast<const float*>(d[0]);
  T* __restrict__ wf = reinterpret_cast<T*>(d[1]);
  T* __restrict__ wd = reinterpret_cast<T*>(d[2]);
  const int Cout = (int)d[3], Cin = (int)d[4], stride2 = (int)d[5];
  const int blk = (int)((long)blockIdx.x - d[6]);
  const int taps = stride2 ? 4 : 16, K = stride2 ? 4 * Cin : Cin;
  if ((Cout % 32) || (Cin % 32)) {
    const long n = (long)((K + 31) / 32) * taps * Cout * 32 + (long)((Cout + 31) / 32) * taps * K * 32;
    const long e0 = (long)blk * 2048;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long e = e0 + i * 256 + threadIdx.x;
      if (e < n) pack4x4_elem<T>(w, wf, wd, Cout, Cin, stride2, e);
    }
    return;
  }
  constexpr int PL = 32 * 33 + 1;                       // plane stride: 1 (mod 32), so the 8 taps a thread scatters hit 8 banks
  __shared__ float tile[8 * PL];
  const int cb = Cin / 32, o0 = (blk / cb) * 32, c0 = (blk % cb) * 32;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {                 // kernel rows kh = 2 pass, 2 pass + 1 (8 of the 16 taps)
    if (pass) __syncthreads();
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
      const int c = idx & 31, o = idx >> 5;
      const float* src = w + ((long)(o0 + o) * Cin + c0 + c) * 16 + pass * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) { tile[t * PL + o * 33 + c] = a[t]; tile[(4 + t) * PL + o * 33 + c] = b[t]; }
    }
    __syncthreads();
    // 8 taps x 32 rows x 4 pieces of 8 = 1024 pieces per operand
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
      const int k8 = (idx & 3) * 8, row = (idx >> 2) & 31, t8 = idx >> 7;
      const int kh = pass * 2 + (t8 >> 2), kw = t8 & 3;
      long fdst, ddst;
      if (stride2) {
        const int ab = (kh >> 1) * 2 + (kw >> 1), rs = (kh & 1) * 2 + (kw & 1);
        const int k0 = rs * Cin + c0;                                  // a multiple of 32
        fdst = (((long)(k0 >> 5) * 4 + ab) * Cout + o0 + row) * 32 + k8;             // wf[chunk][tap][o][kk = c]
        ddst = (((long)(o0 >> 5) * 4 + (3 - ab)) * K + k0 + row) * 32 + k8;          // wd[o chunk][flipped tap][k][oo]
      } else {
        const int t16 = kh * 4 + kw;
        fdst = (((long)(c0 >> 5) * 16 + t16) * Cout + o0 + row) * 32 + k8;
        ddst = (((long)(o0 >> 5) * 16 + (15 - t16)) * K + c0 + row) * 32 + k8;
      }
      f32x8 vf, vd;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vf.v[i] = tile[t8 * PL + row * 33 + k8 + i];                   // o = row, c = k8 + i
        vd.v[i] = tile[t8 * PL + (k8 + i) * 33 + row];                 // c = row, o = k8 + i
      }
      store8(wf + fdst, vf);
      store8(wd + ddst, vd);
    }
  }
}

}  // namespace

extern "C" int s2s_pack_conv4x4(const float* w_oihw, void* wf, void* wd, int Cout, int Cin, int stride, void* stream) {
  return s2s_pack_conv4x4_t(S2S_BF16, w_oihw, wf, wd, Cout, Cin, stride, stream);
}

// wf / wd in the activation dtype (fp32: the parity mode's kernels split the operands into three bf16 on the fly)
extern "C" int s2s_pack_conv4x4_t(int dtype, const float* w_oihw, void* wf, void* wd, int Cout, int Cin, int stride,
                                  void* stream) {
  if (!w_oihw || !wf || !wd) return S2S_ERR_NULL;
  if (Cout <= 0 || Cin <= 0 || (stride != 1 && stride != 2)) return S2S_ERR_SHAPE;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  const int taps = stride == 2 ? 4 : 16, K = stride == 2 ? 4 * Cin : Cin;
  const long n = (long)((K + 31) / 32) * taps * Cout * 32 + (long)((Cout + 31) / 32) * taps * K * 32;
  long nb = (n + 255) / 256;
  if (nb > 65535) nb = 65535;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(pack4x4_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_oihw, (bf16_t*)wf,
                       (bf16_t*)wd, Cout, Cin, stride == 2 ? 1 : 0);
  else
    hipLaunchKernelGGL(pack4x4_kernel<float>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_oihw, (float*)wf,
                       (float*)wd, Cout, Cin, stride == 2 ? 1 : 0);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// blocks a layer occupies in the batched launch (the caller accumulates them into desc[l][6])
extern "C" long s2s_pack_conv4x4_blocks(int Cout, int Cin, int stride) {
  if (Cout <= 0 || Cin <= 0 || (stride != 1 && stride != 2)) return S2S_ERR_SHAPE;
  return pack4x4_layer_blocks(Cout, Cin, stride == 2 ? 1 : 0);
}

// desc: device long[nlayers][7] = {w (fp32 master), wf, wd, Cout, Cin, stride == 2, first block}; total = all blocks
extern "C" int s2s_pack_conv4x4_batched(int dtype, const void* desc, int nlayers, long total, void* stream) {
  if (!desc) return S2S_ERR_NULL;
  if (nlayers <= 0 || total <= 0 || total >= (1L << 31)) return S2S_ERR_SHAPE;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(pack4x4_batched_kernel<bf16_t>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream,
                       (const long*)desc, nlayers);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(pack4x4_batched_kernel<float>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream,
                       (const long*)desc, nlayers);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// ---- per-channel sum over every pixel of an NHWC tensor (conv-bias gradients of the a13 layers) --------------------
// out[c] (+)= sum over [npix] of x[.][c]: the (sum, sum^2) reduction above with the whole batch as one sample, and a
// finalize that keeps the sum.
namespace {

__global__ __launch_bounds__(256) void channel_sum_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* out,
                                                                   int accumulate) {
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= C) return;
  double a, b;
  in_row_sums(part, C, i, nblk, threadIdx.x & 63, a, b);
  if ((threadIdx.x & 63) == 0) out[i] = accumulate ? out[i] + (float)a : (float)a;
}

}  // namespace

extern "C" int s2s_channel_sum_blocks(long npix, int C) {
  if (npix <= 0 || npix >= (1L << 31) || C <= 0 || (C % 8)) return S2S_ERR_SHAPE;
  return in_blocks(1, 1, (int)npix, C);
}

extern "C" int s2s_channel_sum(int dtype, const void* x, int ldx, float* work, float* out, long npix, int C,
                               int accumulate, void* stream) {
  if (!x || !work || !out) return S2S_ERR_NULL;
  if (npix <= 0 || npix >= (1L << 31) || C <= 0 || (C % 8) || (ldx % 8)) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int nb = in_blocks(1, 1, (int)npix, C);
  const dim3 grid(cdiv(C / 8, in_pcb(C)), nb, 1);
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL((in_reduce_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)x, ldx, (const bf16_t*)nullptr,
                       0, (const bf16_t*)nullptr, 0, (const float*)nullptr, work, 1, (int)npix, C, 0.f);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL((in_reduce_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, ldx, (const float*)nullptr, 0,
                       (const float*)nullptr, 0, (const float*)nullptr, work, 1, (int)npix, C, 0.f);
  else return S2S_ERR_DTYPE;
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, work, nb, C, out,
                     accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
