// The two bandwidth-bound convolutions at the ends of the U-Net.
//
//   * stem: Conv3x3(pad 1) from the image tensor (NCHW fp32, Cin <= 6: RGB, or RGB + mask channel) to NHWC features -- the first conv of
//     SharedEncoder.inc (src/models/components/shared_encoder.py:15,67).  Forward: one K = 32 MFMA step on an
//     im2col patch built in LDS (kernel in conv3x3_mfma.hip, next to the shared epilogue); weight/bias gradient:
//     the MFMA kernel below.  The layer is bound by moving its 64-channel output / gradient.
//   * head: Conv1x1 from NHWC features to the NCHW fp32 velocity field -- FlowMatchingDecoder.outc
//     (src/models/components/task_decoders.py:100,132,169), Cout <= 8 (3 RGB, 1 mask, up to 8 classes).
// and their backward passes (weight / bias gradients; the stem needs no data gradient).
#include "common.h"

namespace {

constexpr int STEM_MAX_CIN = 8;

// ---------------------------------------------------------------------------------------------
// stem weight / bias gradient on MFMA:
//   D[co][k] = sum_p dY[p][co] * P[p][k],  k = ci*9 + tap (Cin*9 <= 31 columns used), P[p][Cin*9] = 1 (bias)
// One workgroup walks 16x16 pixel tiles.  dY is staged as [pixel][32 co] images and the im2col patch
// P as a [pixel][32 k] image (built in LDS from the fp32 halo), both read with ds_read_b64_tr_b16 exactly
// like conv3x3_wgrad_mfma.hip.  Wave w owns output channels 32*(w&1).. and the pixel half (w>>1) of
// every tile; partial slabs part[2*gridDim.x][Cout][32] are folded by stem_wgrad_reduce_kernel.
// T = float: three-way bf16 split of both operands, six MFMAs (see conv3x3_mfma.hip).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag8(const char* p) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 64));
  bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = l4[i]; r[4 + i] = h4[i]; }
  return r;
}

template <int NIMG>
__device__ __forceinline__ void split_store(char* base, int img_stride, int off, const float* v8) {
  bf16x8 h, m, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = (bf16_t)v8[i];
    const float r1 = v8[i] - (float)h[i];
    m[i] = (bf16_t)r1;
    l[i] = (bf16_t)(r1 - (float)m[i]);
  }
  *reinterpret_cast<bf16x8*>(base + off) = h;
  if (NIMG == 3) {
    *reinterpret_cast<bf16x8*>(base + img_stride + off) = m;
    *reinterpret_cast<bf16x8*>(base + 2 * img_stride + off) = l;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const T* __restrict__ dy, int lddy,
                                                         const float* __restrict__ x, float* __restrict__ part,
                                                         int B, int H, int W, int Cin, int Cout, int tilesY,
                                                         int tilesX) {
  constexpr int NIMG = sizeof(T) == 4 ? 3 : 1;
  constexpr int NPX = 256;
  constexpr int DY_IMG = 2 * NPX * 64;      // two 32-channel images of 64-B rows
  constexpr int P_IMG = NPX * 64;           // [pixel][32 k]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsDY = smem;                                  // [NIMG][2][256][64B]
  char* const ldsP = smem + NIMG * DY_IMG;                   // [NIMG][256][64B]
  float* const xl = reinterpret_cast<float*>(ldsP + NIMG * P_IMG);  // [Cin][18][18]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave & 1, whalf = wave >> 1;
  const int co0 = blockIdx.y * 64;
  const int c0 = blockIdx.z * 3;                             // channel group (27 of 32 k-columns, +1 for the bias)
  const int nc = (Cin - c0) < 3 ? (Cin - c0) : 3, K = nc * 9;
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + pq * 4) * 2;
  const int ntiles = B * tilesY * tilesX;
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bt = tile;
    const int tx = bt % tilesX; bt /= tilesX;
    const int ty = bt % tilesY;
    const int n = bt / tilesY;
    const int y0 = ty * 16, x0 = tx * 16;
    __syncthreads();   // previous tile fully consumed
    for (int i = tid; i < nc * 324; i += 256) {
      const int ci = i / 324, r = i - ci * 324, hy = r / 18, hx = r - hy * 18;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      xl[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[(((long)n * Cin + c0 + ci) * H + gy) * W + gx] : 0.f;
    }
    // dY tile: 256 px x 64 co = 2048 pieces of 8 channels
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = tid + it * 256;
      const int px = idx >> 3, pc = idx & 7;
      const int gy = y0 + (px >> 4), gx = x0 + (px & 15), co = co0 + pc * 8;
      float v8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v8[k] = 0.f;
      if (gy < H && gx < W && co < Cout) {
        const f32x8 v = load8(dy + (((long)n * H + gy) * W + gx) * lddy + co);
#pragma unroll
        for (int k = 0; k < 8; ++k) v8[k] = v.v[k];
      }
      split_store<NIMG>(ldsDY, DY_IMG, (pc >> 2) * (NPX * 64) + px * 64 + (pc & 3) * 16, v8);
    }
    __syncthreads();   // halo visible
    {                  // im2col row of pixel `tid`: k = ci*9+tap, column K = 1 (bias), rest 0
      const int py = tid >> 4, px = tid & 15;
      const bool inside = (y0 + py < H) && (x0 + px < W);
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        float v8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int kk = c4 * 8 + k;
          float v = 0.f;
          if (kk < K) {
            const int ci = kk / 9, t = kk - ci * 9;
            v = xl[ci * 324 + (py + t / 3) * 18 + px + t % 3];
          } else if (kk == K && c0 == 0) {
            v = inside ? 1.f : 0.f;   // dY is zero outside anyway
          }
          v8[k] = v;
        }
        split_store<NIMG>(ldsP, P_IMG, tid * 64 + c4 * 16, v8);
      }
    }
    __syncthreads();
    const char* A0 = ldsDY + wco * (NPX * 64) + frag_off;
    const char* B0 = ldsP + frag_off;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int m0 = (whalf * 8 + ks) * 16;
      const bf16x8 ah = tr_frag8(A0 + m0 * 64);
      const bf16x8 bh = tr_frag8(B0 + m0 * 64);
      if constexpr (NIMG == 3) {
        const bf16x8 am = tr_frag8(A0 + DY_IMG + m0 * 64), al = tr_frag8(A0 + 2 * DY_IMG + m0 * 64);
        const bf16x8 bm = tr_frag8(B0 + P_IMG + m0 * 64), bl = tr_frag8(B0 + 2 * P_IMG + m0 * 64);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  // D[co][k]: column = lane&31 = k, rows = co
  const int kcol = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int co = co0 + wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
    if (co < Cout)
      part[((((long)blockIdx.z * gridDim.x + blockIdx.x) * 2 + whalf) * Cout + co) * 32 + kcol] = acc[j];
  }
}

// one workgroup per (output channel, channel group): 32 k-columns x 32 slab groups (1024 slabs: 32 loads a thread)
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const float* part, int nslab, int Cout, int Cin, float* dw, float* db,
                                         int accumulate) {
  __shared__ double red[32][33];
  const int c0 = blockIdx.y * 3;
  const int K = ((Cin - c0) < 3 ? (Cin - c0) : 3) * 9;
  const int co = blockIdx.x, k = threadIdx.x & 31, sg = threadIdx.x >> 5;
  part += (long)blockIdx.y * nslab * Cout * 32;
  double s = 0.0, s1 = 0.0;
  int b = sg;
  for (; b + 32 < nslab; b += 64) {
    s += (double)part[((long)b * Cout + co) * 32 + k];
    s1 += (double)part[((long)(b + 32) * Cout + co) * 32 + k];
  }
  for (; b < nslab; b += 32) s += (double)part[((long)b * Cout + co) * 32 + k];
  s += s1;
  red[sg][k] = s;
  __syncthreads();
  if (sg == 0) {
    for (int j = 1; j < 32; ++j) s += red[j][k];
    if (k < K) {
      float* dst = dw + co * Cin * 9 + c0 * 9 + k;
      *dst = accumulate ? *dst + (float)s : (float)s;
    } else if (k == K && c0 == 0 && db) {
      db[co] = accumulate ? db[co] + (float)s : (float)s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// head: 1x1 conv NHWC -> NCHW fp32
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_MAX_COUT = 4;      // fused head+loss kernel (flow head, 3 channels)
constexpr int HEAD_WIDE_COUT = 8;     // plain head kernels (segmentation head: num_classes)

template <typename T, int MO>
__global__ void head_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                const float* __restrict__ bias, float* __restrict__ y, long npix, int HW, int C,
                                int Cout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl = reinterpret_cast<float*>(smem);  // [Cout][C]
  for (int i = threadIdx.x; i < Cout * C; i += blockDim.x) wl[i] = w[i];
  __syncthreads();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    float acc[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) acc[o] = 0.f;
    for (int c8 = 0; c8 < C; c8 += 8) {
      const f32x8 v = load8(x + p * ldx + c8);
#pragma unroll
      for (int o = 0; o < MO; ++o)
        if (o < Cout)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[o] = fmaf(v.v[k], wl[o * C + c8 + k], acc[o]);
    }
    const long n = p / HW, q = p - n * HW;
#pragma unroll
    for (int o = 0; o < MO; ++o)
      if (o < Cout) y[(n * Cout + o) * HW + q] = acc[o] + (bias ? bias[o] : 0.f);
  }
}

// dX[p][c] = sum_o dY[n][o][q] * W[o][c]
template <typename T, int MO>
__global__ void head_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                     int lddx, long npix, int HW, int C, int Cout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < Cout * C; i += blockDim.x) wl[i] = w[i];
  __syncthreads();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long n = p / HW, q = p - n * HW;
    float g[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) g[o] = o < Cout ? dy[(n * Cout + o) * HW + q] : 0.f;
    for (int c8 = 0; c8 < C; c8 += 8) {
      f32x8 v;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < MO; ++o)
          if (o < Cout) a = fmaf(g[o], wl[o * C + c8 + k], a);
        v.v[k] = a;
      }
      store8(dx + p * lddx + c8, v);
    }
  }
}

// dW[o][c] = sum_p dY[p][o]*X[p][c], db[o] = sum_p dY[p][o]; part[gridDim.x][Cout][C+1]
template <typename T, int MO>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dy, const T* __restrict__ x,
                                                         int ldx, float* __restrict__ part, long npix, int HW, int C,
                                                         int Cout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xl = reinterpret_cast<float*>(smem);            // [64 px][C+1]
  float* gl = xl + 64 * (C + 1);                         // [64 px][MO]
  const int nout = Cout * (C + 1);
  float acc[8];                                          // outputs tid, tid+256, ... (nout <= 2048)
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (long p0 = (long)blockIdx.x * 64; p0 < npix; p0 += (long)gridDim.x * 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * C; i += 256) {
      const int pp = i / C, c = i - pp * C;
      xl[pp * (C + 1) + c] = (p0 + pp < npix) ? to_f32(x[(p0 + pp) * ldx + c]) : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * MO; i += 256) {
      const int pp = i / MO, o = i - pp * MO;
      float v = 0.f;
      if (o < Cout && p0 + pp < npix) {
        const long p = p0 + pp, n = p / HW, q = p - n * HW;
        v = dy[(n * Cout + o) * HW + q];
      }
      gl[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = threadIdx.x + i * 256;
      if (id < nout) {
        const int o = id / (C + 1), c = id - o * (C + 1);
        float a = acc[i];
        if (c < C) for (int pp = 0; pp < 64; ++pp) a = fmaf(gl[pp * MO + o], xl[pp * (C + 1) + c], a);
        else for (int pp = 0; pp < 64; ++pp) a += gl[pp * MO + o];
        acc[i] = a;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = threadIdx.x + i * 256;
    if (id < nout) part[(long)blockIdx.x * nout + id] = acc[i];
  }
}

// part[nblk][nout] -> dw/db; 8 outputs x 32 block groups per workgroup (the 195 outputs of the 64 -> 3 head give
// 25 workgroups; with 32 outputs each there were 7 and the 1024-row sum took 34 us)
__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float* part, int nblk, int Cout, int C, float* dw, float* db,
                                         int accumulate) {
  __shared__ double red[32][9];
  const int nout = Cout * (C + 1);
  const int ol = threadIdx.x & 7, sg = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + ol;
  double s = 0.0;
  if (i < nout) {
    double s1 = 0.0;
    int b = sg;
    for (; b + 32 < nblk; b += 64) { s += (double)part[(long)b * nout + i]; s1 += (double)part[(long)(b + 32) * nout + i]; }
    for (; b < nblk; b += 32) s += (double)part[(long)b * nout + i];
    s += s1;
  }
  red[sg][ol] = s;
  __syncthreads();
  if (sg == 0 && i < nout) {
    for (int j = 1; j < 32; ++j) s += red[j][ol];
    const int o = i / (C + 1), c = i - o * (C + 1);
    if (c == C && !db) return;
    float* dst = (c < C) ? dw + o * C + c : db + o;
    *dst = accumulate ? *dst + (float)s : (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// head + loss fused (training step): one pass over the last activation computes
//   v = W a + b,  loss += (v-u)^2,  dv = coef (v-u),  da = W^T dv,  dW += dv a^T,  db += dv
// i.e. FlowMatchingDecoder.outc (task_decoders.py:132) + mean((vt-ut)**2) (conditional_flow_matching.py:72)
// + their backward, reading the 64-channel activation once instead of three times.
// part[gridDim.x][Cout][C+1] (dW | db), lpart[gridDim.x] (double, sum of squares).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_loss_fused_kernel(const T* __restrict__ x, int ldx,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ u, float* __restrict__ v_out,
                                                              T* __restrict__ dx, int lddx, float coef,
                                                              float* __restrict__ part, double* __restrict__ lpart,
                                                              long npix, int HW, int C, int Cout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl = reinterpret_cast<float*>(smem);                 // [Cout][C]
  float* gl = wl + HEAD_MAX_COUT * C;                          // [256][HEAD_MAX_COUT] dv
  bf16_t* al = reinterpret_cast<bf16_t*>(gl + 256 * HEAD_MAX_COUT);   // [256][C+8] activation tile (bf16 is enough for dW)
  __shared__ double lred[256];
  const int AS = C + 8;
  for (int i = threadIdx.x; i < Cout * C; i += 256) wl[i] = w[i];
  const int nout = Cout * (C + 1);
  float acc[4];                                                // outputs tid, tid+256, ... (nout <= 1024)
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = 0.f;
  double lsum = 0.0;
  __syncthreads();
  for (long p0 = (long)blockIdx.x * 256; p0 < npix; p0 += (long)gridDim.x * 256) {
    const long p = p0 + threadIdx.x;
    const bool ok = p < npix;
    float g[HEAD_MAX_COUT];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_COUT; ++o) g[o] = 0.f;
    if (ok) {
      float vv[HEAD_MAX_COUT];
#pragma unroll
      for (int o = 0; o < HEAD_MAX_COUT; ++o) vv[o] = 0.f;
      for (int c8 = 0; c8 < C; c8 += 8) {
        const f32x8 a = load8(x + p * ldx + c8);
        bf16x8 ab;
#pragma unroll
        for (int k = 0; k < 8; ++k) ab[k] = (bf16_t)a.v[k];
        *reinterpret_cast<bf16x8*>(al + threadIdx.x * AS + c8) = ab;
#pragma unroll
        for (int o = 0; o < HEAD_MAX_COUT; ++o)
          if (o < Cout)
#pragma unroll
            for (int k = 0; k < 8; ++k) vv[o] = fmaf(a.v[k], wl[o * C + c8 + k], vv[o]);
      }
      const long n = p / HW, q = p - n * HW;
#pragma unroll
      for (int o = 0; o < HEAD_MAX_COUT; ++o)
        if (o < Cout) {
          const float vo = vv[o] + (bias ? bias[o] : 0.f);
          const long idx = (n * Cout + o) * HW + q;
          if (v_out) v_out[idx] = vo;
          const float d = vo - u[idx];
          lsum += (double)(d * d);
          g[o] = coef * d;
        }
      for (int c8 = 0; c8 < C; c8 += 8) {
        f32x8 da;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float s = 0.f;
#pragma unroll
          for (int o = 0; o < HEAD_MAX_COUT; ++o)
            if (o < Cout) s = fmaf(g[o], wl[o * C + c8 + k], s);
          da.v[k] = s;
        }
        store8(dx + p * lddx + c8, da);
      }
    } else {
      for (int c8 = 0; c8 < C; c8 += 8) {
        bf16x8 z;
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = (bf16_t)0.f;
        *reinterpret_cast<bf16x8*>(al + threadIdx.x * AS + c8) = z;
      }
    }
#pragma unroll
    for (int o = 0; o < HEAD_MAX_COUT; ++o) gl[threadIdx.x * HEAD_MAX_COUT + o] = g[o];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = threadIdx.x + i * 256;
      if (id < nout) {
        const int o = id / (C + 1), c = id - o * (C + 1);
        float a = acc[i];
        if (c < C) for (int pp = 0; pp < 256; ++pp) a = fmaf(gl[pp * HEAD_MAX_COUT + o], (float)al[pp * AS + c], a);
        else for (int pp = 0; pp < 256; ++pp) a += gl[pp * HEAD_MAX_COUT + o];
        acc[i] = a;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + i * 256;
    if (id < nout) part[(long)blockIdx.x * nout + id] = acc[i];
  }
  lred[threadIdx.x] = lsum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) lred[threadIdx.x] += lred[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) lpart[blockIdx.x] = lred[0];
}

// Lane-per-piece form of the fused head (C = 8*PCB, PCB a power of two <= 64): the PCB lanes of a pixel each
// own 8 of its channels, so a wave reads and writes whole NHWC pixel rows (the one-thread-per-pixel form above
// issues 64 different cache lines per load instruction and needs an LDS transpose for dW).  v = W a is finished by
// a butterfly over the PCB lanes; dW and db accumulate in registers over the workgroup's pixels and are folded
// once at the end.  Same outputs and workspace as head_loss_fused_kernel.
template <typename T, int PCB>
__global__ __launch_bounds__(256) void head_loss_lanes_kernel(const T* __restrict__ x, int ldx,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ u, float* __restrict__ v_out,
                                                              T* __restrict__ dx, int lddx, float coef,
                                                              float* __restrict__ part, double* __restrict__ lpart,
                                                              long npix, int HW, int Cout) {
  constexpr int C = PCB * 8, PPB = 256 / PCB, MO = HEAD_MAX_COUT;      // (loops over 3 instead of 4 outputs: no faster)
  __shared__ float fold[256][MO * 8 + 1];
  __shared__ float dbf[PPB][MO];
  __shared__ double lred[256];
  const int tid = threadIdx.x, pc = tid & (PCB - 1), slot = tid / PCB;
  const int lane = tid & 63, grp0 = lane & ~(PCB - 1);
  float wr[MO][8], bo[MO];
#pragma unroll
  for (int o = 0; o < MO; ++o) {
    bo[o] = (o < Cout && bias) ? bias[o] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) wr[o][k] = o < Cout ? w[o * C + pc * 8 + k] : 0.f;
  }
  float dwacc[MO][8], dbacc[MO];
#pragma unroll
  for (int o = 0; o < MO; ++o) {
    dbacc[o] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dwacc[o][k] = 0.f;
  }
  double lsum = 0.0;
  // the next pixel's piece is fetched before this one is processed (one load in flight per thread left the loop
  // latency-bound: 2.4 TB/s); pixel indices fit 32 bits (checked on the host), so n = p / HW is a 32-bit division
  const int np = (int)npix, stride = (int)gridDim.x * PPB;
  f32x8 a_next;
  {
    const int pf = (int)blockIdx.x * PPB + slot;
#pragma unroll
    for (int k = 0; k < 8; ++k) a_next.v[k] = 0.f;
    if (pf < np) a_next = load8(x + (long)pf * ldx + pc * 8);
  }
  for (int p0 = (int)blockIdx.x * PPB; p0 < np; p0 += stride) {
    const int p = p0 + slot;
    const bool ok = p < np;
    const f32x8 a = a_next;
    {
      const int pn = p + stride;
#pragma unroll
      for (int k = 0; k < 8; ++k) a_next.v[k] = 0.f;
      if (p0 + stride < np && pn < np) a_next = load8(x + (long)pn * ldx + pc * 8);
    }
    float vv[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t = fmaf(a.v[k], wr[o][k], t);
#pragma unroll
      for (int m = 1; m < PCB; m <<= 1) t += __shfl_xor(t, m, 64);
      vv[o] = t + bo[o];
    }
    const int n = p / HW, q = p - n * HW;
    float gm[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) {
      gm[o] = 0.f;
      if (ok && o < Cout && (o & (PCB - 1)) == pc) {        // this lane owns output channel o of the pixel
        const long idx = ((long)n * Cout + o) * HW + q;
        if (v_out) v_out[idx] = vv[o];
        const float d = vv[o] - u[idx];
        lsum += (double)(d * d);
        gm[o] = coef * d;
      }
    }
    float g[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) g[o] = __shfl(gm[o], grp0 + (o & (PCB - 1)), 64);
    if (ok) {
      f32x8 da;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = 0.f;
#pragma unroll
        for (int o = 0; o < MO; ++o) t = fmaf(g[o], wr[o][k], t);
        da.v[k] = t;
      }
      store8(dx + (long)p * lddx + pc * 8, da);
    }
#pragma unroll
    for (int o = 0; o < MO; ++o) {
      if (pc == 0) dbacc[o] += g[o];
#pragma unroll
      for (int k = 0; k < 8; ++k) dwacc[o][k] = fmaf(g[o], a.v[k], dwacc[o][k]);
    }
  }
#pragma unroll
  for (int o = 0; o < MO; ++o) {
#pragma unroll
    for (int k = 0; k < 8; ++k) fold[tid][o * 8 + k] = dwacc[o][k];
    if (pc == 0) dbf[slot][o] = dbacc[o];
  }
  lred[tid] = lsum;
  __syncthreads();
  const int nout = Cout * (C + 1);
  for (int id = tid; id < nout; id += 256) {
    const int o = id / (C + 1), c = id - o * (C + 1);
    float t = 0.f;
    if (c < C) {
      const int cp = c >> 3, k = c & 7;
      for (int sl = 0; sl < PPB; ++sl) t += fold[sl * PCB + cp][o * 8 + k];
    } else {
      for (int sl = 0; sl < PPB; ++sl) t += dbf[sl][o];
    }
    part[(long)blockIdx.x * nout + id] = t;
  }
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) lred[tid] += lred[tid + o];
    __syncthreads();
  }
  if (tid == 0) lpart[blockIdx.x] = lred[0];
}

// The same lane-per-piece layout for the module path, where the loss lives outside (autograd drives the head alone):
// FWD = v = W a + b only; otherwise the backward from an incoming dY (NCHW fp32): da = W^T dy, dW += dy a^T, db += dy.
// (The one-thread-per-pixel kernels above read 64 different cache lines per load instruction: 122 / 100 / 348 us for the
// forward / data gradient / weight gradient of the 64 -> 3 head at 16 x 256 x 256, against ~35 us of HBM time each.)
template <typename T, int PCB, bool FWD>
__global__ __launch_bounds__(256) void head_lanes_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ dy,
                                                         float* __restrict__ v_out, T* __restrict__ dx, int lddx,
                                                         float* __restrict__ part, long npix, int HW, int Cout) {
  constexpr int C = PCB * 8, PPB = 256 / PCB, MO = HEAD_MAX_COUT;
  __shared__ float fold[FWD ? 1 : 256][MO * 8 + 1];
  __shared__ float dbf[PPB][MO];
  const int tid = threadIdx.x, pc = tid & (PCB - 1), slot = tid / PCB;
  const int lane = tid & 63, grp0 = lane & ~(PCB - 1);
  float wr[MO][8], bo[MO];
#pragma unroll
  for (int o = 0; o < MO; ++o) {
    bo[o] = (o < Cout && bias) ? bias[o] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) wr[o][k] = o < Cout ? w[o * C + pc * 8 + k] : 0.f;
  }
  float dwacc[MO][8], dbacc[MO];
#pragma unroll
  for (int o = 0; o < MO; ++o) {
    dbacc[o] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dwacc[o][k] = 0.f;
  }
  const int np = (int)npix, stride = (int)gridDim.x * PPB;
  f32x8 a_next;
  {
    const int pf = (int)blockIdx.x * PPB + slot;
#pragma unroll
    for (int k = 0; k < 8; ++k) a_next.v[k] = 0.f;
    if (pf < np) a_next = load8(x + (long)pf * ldx + pc * 8);
  }
  for (int p0 = (int)blockIdx.x * PPB; p0 < np; p0 += stride) {
    const int p = p0 + slot;
    const bool ok = p < np;
    const f32x8 a = a_next;
    {
      const int pn = p + stride;
#pragma unroll
      for (int k = 0; k < 8; ++k) a_next.v[k] = 0.f;
      if (p0 + stride < np && pn < np) a_next = load8(x + (long)pn * ldx + pc * 8);
    }
    const int n = p / HW, q = p - n * HW;
    if (FWD) {
#pragma unroll
      for (int o = 0; o < MO; ++o) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t = fmaf(a.v[k], wr[o][k], t);
#pragma unroll
        for (int m = 1; m < PCB; m <<= 1) t += __shfl_xor(t, m, 64);
        if (ok && o < Cout && (o & (PCB - 1)) == pc) v_out[((long)n * Cout + o) * HW + q] = t + bo[o];
      }
    } else {
      float gm[MO], g[MO];
#pragma unroll
      for (int o = 0; o < MO; ++o) {
        gm[o] = 0.f;
        if (ok && o < Cout && (o & (PCB - 1)) == pc) gm[o] = dy[((long)n * Cout + o) * HW + q];
      }
#pragma unroll
      for (int o = 0; o < MO; ++o) g[o] = __shfl(gm[o], grp0 + (o & (PCB - 1)), 64);
      if (ok) {
        f32x8 da;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float t = 0.f;
#pragma unroll
          for (int o = 0; o < MO; ++o) t = fmaf(g[o], wr[o][k], t);
          da.v[k] = t;
        }
        store8(dx + (long)p * lddx + pc * 8, da);
      }
#pragma unroll
      for (int o = 0; o < MO; ++o) {
        if (pc == 0) dbacc[o] += g[o];
#pragma unroll
        for (int k = 0; k < 8; ++k) dwacc[o][k] = fmaf(g[o], a.v[k], dwacc[o][k]);
      }
    }
  }
  if (FWD) return;
#pragma unroll
  for (int o = 0; o < MO; ++o) {
#pragma unroll
    for (int k = 0; k < 8; ++k) fold[tid][o * 8 + k] = dwacc[o][k];
    if (pc == 0) dbf[slot][o] = dbacc[o];
  }
  __syncthreads();
  const int nout = Cout * (C + 1);
  for (int id = tid; id < nout; id += 256) {
    const int o = id / (C + 1), c = id - o * (C + 1);
    float t = 0.f;
    if (c < C) {
      const int cp = c >> 3, k = c & 7;
      for (int sl = 0; sl < PPB; ++sl) t += fold[sl * PCB + cp][o * 8 + k];
    } else {
      for (int sl = 0; sl < PPB; ++sl) t += dbf[sl][o];
    }
    part[(long)blockIdx.x * nout + id] = t;
  }
}

// launch head_lanes_kernel for C = 8 PCB (PCB a power of two <= 64) and Cout <= 4; false when the shape is not taken
template <typename T, bool FWD>
bool launch_head_lanes(const void* x, int ldx, const float* w, const float* bias, const float* dy, float* v, void* dx,
                       int lddx, float* part, int nb, long npix, int HW, int C, int Cout, hipStream_t s) {
  if (Cout > HEAD_MAX_COUT || (C % 8) || npix >= (1L << 31) - (1L << 20)) return false;
#define S2S_HLN(PP)                                                                                                  \
  hipLaunchKernelGGL((head_lanes_kernel<T, PP, FWD>), dim3(nb), dim3(256), 0, s, (const T*)x, ldx, w, bias, dy, v,   \
                     (T*)dx, lddx, part, npix, HW, Cout);                                                            \
  return true;
  switch (C / 8) {
    case 1: S2S_HLN(1) case 2: S2S_HLN(2) case 4: S2S_HLN(4) case 8: S2S_HLN(8) case 16: S2S_HLN(16) case 32: S2S_HLN(32)
    case 64: S2S_HLN(64) default: return false;
  }
#undef S2S_HLN
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* part, int n, double inv_count, float* loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)(red[0] * inv_count);
}

}  // namespace

// MFMA stem forward lives next to the shared epilogue in conv3x3_mfma.hip
int s2s_internal_stem_fwd(int dtype, const float* x_nchw, const float* w_oihw, const float* bias, void* y, int ldy,
                          float* stat_part, int B, int H, int W, int Cin, int Cout, hipStream_t s);

extern "C" int s2s_stem_stat_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return S2S_ERR_SHAPE;
  return B * cdiv(H, 16) * cdiv(W, 16);
}

extern "C" int s2s_stem_conv3x3_fwd(int dtype, const float* x_nchw, const float* w_oihw, const float* bias, void* y,
                                    int ldy, float* stat_part, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!x_nchw || !w_oihw || !y) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin > 6 || Cout <= 0 || (Cout % 8) || (ldy % 8))
    return S2S_ERR_SHAPE;
  return s2s_internal_stem_fwd(dtype, x_nchw, w_oihw, bias, y, ldy, stat_part, B, H, W, Cin, Cout,
                               (hipStream_t)stream);
}

extern "C" int s2s_stem_wgrad_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return S2S_ERR_SHAPE;
  const int nt = B * cdiv(H, 16) * cdiv(W, 16);
  constexpr int cap = 512;   // 2 per CU: 98 us vs 126 us at 256
  return nt < cap ? nt : cap;
}

// part: float[ceil(Cin/3)][2*blocks][Cout][32]
extern "C" int s2s_stem_conv3x3_wgrad(int dtype, const void* dy, int lddy, const float* x_nchw, float* part,
                                      float* dw_oihw, float* dbias, int accumulate, int B, int H, int W, int Cin,
                                      int Cout, void* stream) {
  if (!dy || !x_nchw || !part || !dw_oihw) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin > 6 || Cout <= 0 || (Cout % 8) || (lddy % 8))
    return S2S_ERR_SHAPE;
  const int nb = s2s_stem_wgrad_blocks(B, H, W);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(nb, cdiv(Cout, 64), cdiv(Cin, 3));
  if (dtype == S2S_BF16) {
    const int lds = 1 * (2 * 256 * 64 + 256 * 64) + 3 * 324 * 4;
    hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)dy, lddy, x_nchw, part, B,
                       H, W, Cin, Cout, cdiv(H, 16), cdiv(W, 16));
  } else if (dtype == S2S_F32) {
    const int lds = 3 * (2 * 256 * 64 + 256 * 64) + 3 * 324 * 4;
    static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
    if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(stem_wgrad_kernel<float>), lds, &attr_devs)) return rc;
    hipLaunchKernelGGL(stem_wgrad_kernel<float>, grid, dim3(256), lds, s, (const float*)dy, lddy, x_nchw, part, B, H,
                       W, Cin, Cout, cdiv(H, 16), cdiv(W, 16));
  } else return S2S_ERR_DTYPE;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(Cout, cdiv(Cin, 3)), dim3(1024), 0, s, part, 2 * nb, Cout, Cin, dw_oihw, dbias,
                     accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_head_conv1x1_fwd(int dtype, const void* x, int ldx, const float* w, const float* bias,
                                    float* y_nchw, int B, int H, int W, int C, int Cout, void* stream) {
  if (!x || !w || !y_nchw) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldx % 8) || Cout <= 0 || Cout > HEAD_WIDE_COUT)
    return S2S_ERR_SHAPE;
  const long npix = (long)B * H * W;
  long grid = (npix + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  {
    const int nbl = (int)(((npix + 255) / 256) < 2048 ? ((npix + 255) / 256) : 2048);
    const bool done = dtype == S2S_BF16
        ? launch_head_lanes<bf16_t, true>(x, ldx, w, bias, nullptr, y_nchw, nullptr, 8, nullptr, nbl, npix, H * W, C, Cout, s)
        : launch_head_lanes<float, true>(x, ldx, w, bias, nullptr, y_nchw, nullptr, 8, nullptr, nbl, npix, H * W, C, Cout, s);
    if (done) { S2S_LAUNCH_CHECK(); return S2S_OK; }
  }
#define S2S_HEAD_FWD(TT, MO)                                                                                        \
  hipLaunchKernelGGL((head_fwd_kernel<TT, MO>), dim3((int)grid), dim3(256), Cout * C * 4, s, (const TT*)x, ldx, w, \
                     bias, y_nchw, npix, H * W, C, Cout)
  if (dtype == S2S_BF16) { if (Cout <= 4) S2S_HEAD_FWD(bf16_t, 4); else S2S_HEAD_FWD(bf16_t, 8); }
  else if (dtype == S2S_F32) { if (Cout <= 4) S2S_HEAD_FWD(float, 4); else S2S_HEAD_FWD(float, 8); }
  else return S2S_ERR_DTYPE;
#undef S2S_HEAD_FWD
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_head_wgrad_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return S2S_ERR_SHAPE;
  const long nb = ((long)B * H * W + 63) / 64;
  return (int)(nb < 512 ? nb : 512);
}

// part: float[blocks][Cout][C+1]
extern "C" int s2s_head_conv1x1_bwd(int dtype, const float* dy_nchw, const void* x, int ldx, const float* w, void* dx,
                                    int lddx, float* part, float* dw, float* dbias, int accumulate, int B, int H, int W,
                                    int C, int Cout, void* stream) {
  if (!dy_nchw || !x || !w || !dx || !part || !dw) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (lddx % 8) || Cout <= 0 || Cout > HEAD_WIDE_COUT)
    return S2S_ERR_SHAPE;
  if (Cout * (C + 1) > 2048) return S2S_ERR_SHAPE;
  const long npix = (long)B * H * W;
  long grid = (npix + 255) / 256;
  if (grid > 4096) grid = 4096;
  const int nb = s2s_head_wgrad_blocks(B, H, W);
  const int mo = Cout <= 4 ? 4 : 8;
  const int lds = (64 * (C + 1) + 64 * mo) * 4;
  hipStream_t s = (hipStream_t)stream;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  {
    // one pass over the activation for both gradients (lane-per-piece form, partial dW | db rows per workgroup)
    const bool done = dtype == S2S_BF16
        ? launch_head_lanes<bf16_t, false>(x, ldx, w, nullptr, dy_nchw, nullptr, dx, lddx, part, nb, npix, H * W, C, Cout, s)
        : launch_head_lanes<float, false>(x, ldx, w, nullptr, dy_nchw, nullptr, dx, lddx, part, nb, npix, H * W, C, Cout, s);
    if (done) {
      hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(Cout * (C + 1), 8)), dim3(256), 0, s, part, nb, Cout, C,
                         dw, dbias, accumulate);
      S2S_LAUNCH_CHECK();
      return S2S_OK;
    }
  }
#define S2S_HEAD_BWD(TT, MO)                                                                                       \
  hipLaunchKernelGGL((head_bwd_data_kernel<TT, MO>), dim3((int)grid), dim3(256), Cout * C * 4, s, dy_nchw, w,      \
                     (TT*)dx, lddx, npix, H * W, C, Cout);                                                         \
  hipLaunchKernelGGL((head_wgrad_kernel<TT, MO>), dim3(nb), dim3(256), lds, s, dy_nchw, (const TT*)x, ldx, part,   \
                     npix, H * W, C, Cout);
  if (dtype == S2S_BF16) { if (mo == 4) { S2S_HEAD_BWD(bf16_t, 4) } else { S2S_HEAD_BWD(bf16_t, 8) } }
  else if (dtype == S2S_F32) { if (mo == 4) { S2S_HEAD_BWD(float, 4) } else { S2S_HEAD_BWD(float, 8) } }
  else return S2S_ERR_DTYPE;
#undef S2S_HEAD_BWD
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(Cout * (C + 1), 8)), dim3(256), 0, s, part, nb, Cout, C,
                     dw, dbias, accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_head_loss_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return S2S_ERR_SHAPE;
  const long nb = ((long)B * H * W + 255) / 256;
  return (int)(nb < 1024 ? nb : 1024);
}

// part: float[blocks][Cout][C+1]; lpart: double[blocks]; v_out (optional) NCHW fp32; dv scale = 2*grad_scale/count
extern "C" int s2s_head_loss_fused(int dtype, const void* x, int ldx, const float* w, const float* bias,
                                   const float* u_nchw, float* v_nchw, void* dx, int lddx, float grad_scale,
                                   float* part, double* lpart, float* dw, float* dbias, float* loss, int accumulate,
                                   int B, int H, int W, int C, int Cout, void* stream) {
  if (!x || !w || !u_nchw || !dx || !part || !lpart || !dw || !loss) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (lddx % 8) || Cout <= 0 || Cout > HEAD_MAX_COUT)
    return S2S_ERR_SHAPE;
  if (Cout * (C + 1) > 1024) return S2S_ERR_SHAPE;
  const long npix = (long)B * H * W;
  if (npix >= (1L << 31) - (1L << 20)) return S2S_ERR_SHAPE;       // 32-bit pixel index (+ one grid stride) in the kernel
  const double count = (double)npix * Cout;
  const int nb = s2s_head_loss_blocks(B, H, W);
  const int lds = (HEAD_MAX_COUT * C + 256 * HEAD_MAX_COUT) * 4 + 256 * (C + 8) * 2;
  if (lds > 60 * 1024) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const float coef = (float)(2.0 * (double)grad_scale / count);
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  const int pcb = C / 8;
  bool done = false;
#define S2S_HL(TT, PP)                                                                                              \
  hipLaunchKernelGGL((head_loss_lanes_kernel<TT, PP>), dim3(nb), dim3(256), 0, s, (const TT*)x, ldx, w, bias, u_nchw, \
                     v_nchw, (TT*)dx, lddx, coef, part, lpart, npix, H * W, Cout);                                 \
  done = true;
#define S2S_HL_T(TT)                                                                                    \
  switch (pcb) {                                                                                        \
    case 1: { S2S_HL(TT, 1) } break;  case 2: { S2S_HL(TT, 2) } break;  case 4: { S2S_HL(TT, 4) } break;    \
    case 8: { S2S_HL(TT, 8) } break;  case 16: { S2S_HL(TT, 16) } break; case 32: { S2S_HL(TT, 32) } break;  \
    case 64: { S2S_HL(TT, 64) } break; default: break;                                                  \
  }
  if (dtype == S2S_BF16) { S2S_HL_T(bf16_t) } else { S2S_HL_T(float) }
#undef S2S_HL_T
#undef S2S_HL
  if (!done) {
    if (dtype == S2S_BF16)
      hipLaunchKernelGGL(head_loss_fused_kernel<bf16_t>, dim3(nb), dim3(256), lds, s, (const bf16_t*)x, ldx, w, bias,
                         u_nchw, v_nchw, (bf16_t*)dx, lddx, coef, part, lpart, npix, H * W, C, Cout);
    else
      hipLaunchKernelGGL(head_loss_fused_kernel<float>, dim3(nb), dim3(256), lds, s, (const float*)x, ldx, w, bias,
                         u_nchw, v_nchw, (float*)dx, lddx, coef, part, lpart, npix, H * W, C, Cout);
  }
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(Cout * (C + 1), 8)), dim3(256), 0, s, part, nb, Cout, C,
                     dw, dbias, accumulate);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, lpart, nb, 1.0 / count, loss);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
