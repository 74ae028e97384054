// Bilinear x2 up-sampling with align_corners=True (+ the zero pad to the skip's size), NHWC.
//
// Replaces nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) followed by F.pad in
// the reference's Up block (src/models/components/task_decoders.py:34,42-47) and its backward.
// The forward writes straight into a channel slice of the decoder's concat buffer (explicit pixel
// stride), optionally adding the per-(sample, channel) time-conditioning bias to its input first
// (FlowMatchingDecoder.forward: x = bottleneck + t[:, :, None, None], task_decoders.py:119-125).
//
// Index arithmetic follows ATen's upsample_bilinear2d: scale = (in-1)/(out-1) in fp32,
// src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), w1 = src - i0, w0 = 1 - w1.
#include "common.h"
#include <stdlib.h>

namespace {

struct Axis {
  int i0, i1;
  float w0, w1;
};

__device__ __forceinline__ Axis ac_axis(int dst, int n_in, float scale) {
  Axis a;
  const float src = scale * (float)dst;
  a.i0 = (int)src;
  if (a.i0 > n_in - 1) a.i0 = n_in - 1;
  a.i1 = a.i0 + (a.i0 < n_in - 1 ? 1 : 0);
  a.w1 = src - (float)a.i0;
  a.w0 = 1.f - a.w1;
  return a;
}

// Thread layout (both directions): a workgroup owns one image row; lanes walk the channel pieces of a pixel first
// (PCB pieces of 8 channels, so a wave touches whole NHWC pixel rows) and the remaining 256/PCB thread slots walk
// the pixels of the row.  Row quantities (vertical tap and weights, row base pointers) are wave-uniform and there
// is no per-element div/mod; the earlier flat-index kernels spent ~300 VALU instructions per 16 B and ran at a
// third of the HBM rate.
__device__ __forceinline__ int rs_pcb(int C) {
  int p = 1;
  while (p < 32 && p * 8 < C) p <<= 1;
  return p;
}

template <typename T, bool BIAS>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, int ldx,
                                                             const float* __restrict__ bias, T* __restrict__ y,
                                                             int ldy, int B, int Hin, int Win, int Hout, int Wout,
                                                             int padT, int padL, int C, int band) {
  const int cp = C >> 3;
  const int Hu = 2 * Hin, Wu = 2 * Win;
  const float sh = Hu > 1 ? (float)(Hin - 1) / (float)(Hu - 1) : 0.f;
  const float sw = Wu > 1 ? (float)(Win - 1) / (float)(Wu - 1) : 0.f;
  const int PCB = rs_pcb(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), slot = threadIdx.x / PCB;
  // A workgroup takes BANDS of `band` consecutive output rows (round 3): consecutive output rows read the same two
  // input rows, and workgroups are dealt to the eight XCDs round-robin, so with one row per workgroup every XCD's L2
  // fetched every input row (the counters had the launch at 1.6x its algorithmic bytes); a band keeps the shared rows
  // in one L2.
  const int nrows = B * Hout;
  for (int row0 = blockIdx.x * band; row0 < nrows; row0 += gridDim.x * band)
  for (int row = row0; row < row0 + band && row < nrows; ++row) {
    const int n = row / Hout, oy = row - n * Hout;
    const int uy = oy - padT;
    const bool row_in = uy >= 0 && uy < Hu;
    const Axis ay = ac_axis(row_in ? uy : 0, Hin, sh);
    const T* const r0 = x + ((long)n * Hin + ay.i0) * Win * ldx;
    const T* const r1 = x + ((long)n * Hin + ay.i1) * Win * ldx;
    T* const yr = y + (long)row * Wout * ldy;
    for (int pb = 0; pb < cp; pb += PCB) {
      const int c8 = (pb + pc) * 8;
      if (pb + pc >= cp) continue;
      float bv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) bv[k] = BIAS ? bias[(long)n * C + c8 + k] : 0.f;
      for (int ox = slot; ox < Wout; ox += WL) {
        const int ux = ox - padL;
        f32x8 o;
        if (!row_in || ux < 0 || ux >= Wu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
        } else {
          const Axis ax = ac_axis(ux, Win, sw);
          const int o0 = ax.i0 * ldx + c8, o1 = ax.i1 * ldx + c8;
          const f32x8 v00 = load8(r0 + o0), v01 = load8(r0 + o1), v10 = load8(r1 + o0), v11 = load8(r1 + o1);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float b = bv[k];
            o.v[k] = ay.w0 * (ax.w0 * (v00.v[k] + b) + ax.w1 * (v01.v[k] + b)) +
                     ay.w1 * (ax.w0 * (v10.v[k] + b) + ax.w1 * (v11.v[k] + b));
          }
        }
        store8(yr + ox * ldy + c8, o);
      }
    }
  }
}

// Gather form of the backward: input pixel (iy, ix) collects from every up-sampled pixel that read
// it.  With scale < 1 at most 3 destination rows map their i0 or i1 onto one source row; the
// candidates are bracketed from the inverse scale and tested exactly with the forward's own index
// arithmetic, so forward and backward can never disagree.
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int B,
                                      int Hin, int Win, int Hout, int Wout, int padT, int padL, int C) {
  const int cp = C >> 3;
  const int Hu = 2 * Hin, Wu = 2 * Win;
  const float sh = Hu > 1 ? (float)(Hin - 1) / (float)(Hu - 1) : 0.f;
  const float sw = Wu > 1 ? (float)(Win - 1) / (float)(Wu - 1) : 0.f;
  const int total = B * Hin * Win * cp;             // 32-bit element index (checked on the host)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t0 = i / cp, c8 = (i - t0 * cp) * 8;
    const int t1 = t0 / Win, ix = t0 - t1 * Win;
    const int n = t1 / Hin, iy = t1 - n * Hin;
    // candidate destination range: rows whose src coordinate lies in (iy-1, iy+1)
    int y_lo, y_hi, x_lo, x_hi;
    if (Hin == 1) { y_lo = 0; y_hi = Hu - 1; }
    else {
      y_lo = (int)floorf((float)(iy - 1) / sh) - 1; y_hi = (int)ceilf((float)(iy + 1) / sh) + 1;
      if (y_lo < 0) y_lo = 0;
      if (y_hi > Hu - 1) y_hi = Hu - 1;
    }
    if (Win == 1) { x_lo = 0; x_hi = Wu - 1; }
    else {
      x_lo = (int)floorf((float)(ix - 1) / sw) - 1; x_hi = (int)ceilf((float)(ix + 1) / sw) + 1;
      if (x_lo < 0) x_lo = 0;
      if (x_hi > Wu - 1) x_hi = Wu - 1;
    }
    // column candidates once per source pixel (they were re-derived for every contributing row): slot j = ux - x_lo,
    // weight 0 = does not read this source column / outside the padded output
    constexpr int NSLOT = 12;                         // the bracket holds at most 2/scale + 5 <= 11 candidates
    float wxs[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int ux = x_lo + j;
      float wx = 0.f;
      if (ux <= x_hi) {
        const Axis ax = ac_axis(ux, Win, sw);
        if (ax.i0 == ix) wx += ax.w0;
        if (ax.i1 == ix) wx += ax.w1;
        const int ox = ux + padL;
        if (ox < 0 || ox >= Wout) wx = 0.f;
      }
      wxs[j] = wx;
    }
    f32x8 acc;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
    for (int uy = y_lo; uy <= y_hi; ++uy) {
      const Axis ay = ac_axis(uy, Hin, sh);
      float wy = 0.f;
      if (ay.i0 == iy) wy += ay.w0;
      if (ay.i1 == iy) wy += ay.w1;
      const int oy = uy + padT;
      if (wy == 0.f || oy < 0 || oy >= Hout) continue;
      const T* const rowp = dy + (((long)n * Hout + oy) * Wout + x_lo + padL) * lddy + c8;
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
        if (wxs[j] == 0.f) continue;
        const f32x8 g = load8(rowp + (long)j * lddy);
        const float w = wy * wxs[j];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc.v[k] += w * g.v[k];
      }
    }
    store8(dx + (((long)n * Hin + iy) * Win + ix) * lddx + c8, acc);
  }
}

// The same gather with a FIXED 5 x 5 window of unconditional loads (maps wider than 3 source pixels per axis).  The
// rows that read source row iy are those whose i0 is iy - 1 or iy: consecutive, and at most five of them when
// 2 (2 Hin - 1) / (Hin - 1) < 5, i.e. Hin > 3.  The first one is found by walking the forward's own index arithmetic up
// from a conservative estimate (no loads), the five weights follow from it (zero where a row does not contribute or
// falls outside), and the 25 loads of a thread are independent of any weight test, so they are all in flight at once;
// the loop above issues each load behind a data-dependent branch and is latency-bound (2.3 TB/s).
__device__ __forceinline__ float ac_weight_of(int dst, int n_in, float scale, int src_index) {
  const Axis a = ac_axis(dst, n_in, scale);
  return (a.i0 == src_index ? a.w0 : 0.f) + (a.i1 == src_index ? a.w1 : 0.f);
}

template <typename T>
__global__ void upsample2x_bwd_win_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int B,
                                          int Hin, int Win, int Hout, int Wout, int padT, int padL, int C) {
  const int cp = C >> 3;
  const int Hu = 2 * Hin, Wu = 2 * Win;
  const float sh = (float)(Hin - 1) / (float)(Hu - 1), sw = (float)(Win - 1) / (float)(Wu - 1);
  const int total = B * Hin * Win * cp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t0 = i / cp, c8 = (i - t0 * cp) * 8;
    const int t1 = t0 / Win, ix = t0 - t1 * Win;
    const int n = t1 / Hin, iy = t1 - n * Hin;
    int uy0 = (int)floorf((float)(iy - 1) / sh) - 1, ux0 = (int)floorf((float)(ix - 1) / sw) - 1;
    if (uy0 < 0) uy0 = 0;
    if (ux0 < 0) ux0 = 0;
    for (int t = 0; t < 8 && uy0 < Hu - 1 && ac_weight_of(uy0, Hin, sh, iy) == 0.f; ++t) ++uy0;
    for (int t = 0; t < 8 && ux0 < Wu - 1 && ac_weight_of(ux0, Win, sw, ix) == 0.f; ++t) ++ux0;
    float wy[5], wx[5];
    int oyc[5], oxc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int uy = uy0 + j, ux = ux0 + j;
      const int oy = uy + padT, ox = ux + padL;
      wy[j] = (uy <= Hu - 1 && oy >= 0 && oy < Hout) ? ac_weight_of(uy, Hin, sh, iy) : 0.f;
      wx[j] = (ux <= Wu - 1 && ox >= 0 && ox < Wout) ? ac_weight_of(ux, Win, sw, ix) : 0.f;
      oyc[j] = oy < 0 ? 0 : (oy > Hout - 1 ? Hout - 1 : oy);
      oxc[j] = ox < 0 ? 0 : (ox > Wout - 1 ? Wout - 1 : ox);
    }
    f32x8 acc;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc.v[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const T* const rowp = dy + ((long)n * Hout + oyc[j]) * Wout * lddy + c8;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const f32x8 g = load8(rowp + (long)oxc[m] * lddy);
        const float w = wy[j] * wx[m];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc.v[k] = fmaf(w, g.v[k], acc.v[k]);
      }
    }
    store8(dx + (((long)n * Hin + iy) * Win + ix) * lddx + c8, acc);
  }
}

// out[n][c] (+)= sum over the H*W pixels of x[n][.][.][c]   (gradient of the broadcast time bias)
template <typename T>
__global__ void pixel_sum_kernel(const T* __restrict__ x, int ldx, float* __restrict__ out, int HW, int C,
                                 int accumulate) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;  // 4 pixel groups
  float s = 0.f;
  if (c < C)
    for (int p = rg; p < HW; p += 4) s += to_f32(x[((long)n * HW + p) * ldx + c]);
  __shared__ float red[4][64];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    out[(long)n * C + c] = accumulate ? out[(long)n * C + c] + s : s;
  }
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int s2s_upsample2x_bilinear_ac_fwd(int dtype, const void* x, int ldx, const float* bias_nc, void* y,
                                              int ldy, int B, int Hin, int Win, int Hout, int Wout, int C,
                                              void* stream) {
  if (!x || !y) return S2S_ERR_NULL;
  if (B <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (ldy % 8)) return S2S_ERR_SHAPE;
  if (Hout < 2 * Hin || Wout < 2 * Win) return S2S_ERR_SHAPE;  // F.pad with a negative size (crop) is not supported
  const int padT = (Hout - 2 * Hin) / 2, padL = (Wout - 2 * Win) / 2;
  if ((long)Win * ldx >= (1L << 31) || (long)Wout * ldy >= (1L << 31) || (long)B * Hout >= (1L << 31))
    return S2S_ERR_SHAPE;                                        // 32-bit offsets inside a row
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)B * Hout;
  // rows per workgroup: 4 while that still leaves >= 1024 workgroups (4 per CU), else 2, else 1 (S2S_UP_BAND overrides)
  static const int band_env = [] { const char* e = getenv("S2S_UP_BAND"); return e ? atoi(e) : 0; }();
  const int band = band_env > 0 ? band_env : (rows >= 4096 ? 4 : rows >= 2048 ? 2 : 1);
  const long nwg = (rows + band - 1) / band;
  const dim3 grid((unsigned)(nwg < 65535 * 4 ? nwg : 65535 * 4));
#define S2S_UP(TT, BB)                                                                                          \
  hipLaunchKernelGGL((upsample2x_fwd_kernel<TT, BB>), grid, dim3(256), 0, s, (const TT*)x, ldx, bias_nc, (TT*)y, \
                     ldy, B, Hin, Win, Hout, Wout, padT, padL, C, band)
  if (dtype == S2S_BF16) { if (bias_nc) S2S_UP(bf16_t, true); else S2S_UP(bf16_t, false); }
  else if (dtype == S2S_F32) { if (bias_nc) S2S_UP(float, true); else S2S_UP(float, false); }
  else return S2S_ERR_DTYPE;
#undef S2S_UP
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_upsample2x_bilinear_ac_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx, int B, int Hin,
                                              int Win, int Hout, int Wout, int C, void* stream) {
  if (!dy || !dx) return S2S_ERR_NULL;
  if (B <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || (C % 8) || (lddy % 8) || (lddx % 8)) return S2S_ERR_SHAPE;
  if (Hout < 2 * Hin || Wout < 2 * Win) return S2S_ERR_SHAPE;
  const int padT = (Hout - 2 * Hin) / 2, padL = (Wout - 2 * Win) / 2;
  const long total = (long)B * Hin * Win * (C / 8);
  if (total >= (1L << 31) - (1L << 22)) return S2S_ERR_SHAPE;   // 32-bit element index in the kernel
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(ew_grid(total));
  const bool win = Hin > 3 && Win > 3;       // at most five contributing rows / columns per source pixel (else the branching gather)
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
#define S2S_UPB(KERN, TT)                                                                                            \
  hipLaunchKernelGGL(KERN<TT>, grid, dim3(256), 0, s, (const TT*)dy, lddy, (TT*)dx, lddx, B, Hin, Win, Hout, Wout,   \
                     padT, padL, C)
  if (dtype == S2S_BF16) { if (win) S2S_UPB(upsample2x_bwd_win_kernel, bf16_t); else S2S_UPB(upsample2x_bwd_kernel, bf16_t); }
  else { if (win) S2S_UPB(upsample2x_bwd_win_kernel, float); else S2S_UPB(upsample2x_bwd_kernel, float); }
#undef S2S_UPB
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_pixel_sum(int dtype, const void* x, int ldx, float* out_nc, int B, int HW, int C, int accumulate,
                             void* stream) {
  if (!x || !out_nc) return S2S_ERR_NULL;
  if (B <= 0 || HW <= 0 || C <= 0) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(C, 64), B);
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(pixel_sum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, out_nc, HW, C, accumulate);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(pixel_sum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, out_nc, HW, C, accumulate);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
