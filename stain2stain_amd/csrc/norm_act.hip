// BatchNorm2d (training / eval) + ReLU (+ MaxPool2d(2)) around the convolutions, NHWC, HBM-bound.
//
// Replaces nn.BatchNorm2d -> nn.ReLU(inplace) (-> nn.MaxPool2d(2) of the next Down block) of the
// reference's DoubleConv / Down (src/models/components/shared_encoder.py:16-20,32-37;
// task_decoders.py:16-20) and their autograd backward.
//
//   forward : conv epilogue leaves per-row-block partial (sum, sum^2) -> s2s_bn_finalize
//             (mean, 1/std, folded scale/shift, running-stat update) -> s2s_bn_relu_apply
//             (one pass: y = relu(x*scale+shift), optionally also the 2x2 max-pooled y)
//   backward: s2s_bn_relu_bwd_reduce (sum dz, sum dz*xhat; dz = g*(y>0); g may include the
//             max-pool scatter of the next level's gradient, recomputed from y) ->
//             s2s_bn_bwd_finalize -> s2s_bn_relu_bwd_apply (dx, + partial sum of dx = conv-bias grad)
//
// All tensors are [B][H][W][C] views with an explicit pixel stride ("ld", in elements) so that a
// channel slice of a wider buffer (the decoder's concat buffer) can be read or written in place.
// Every thread moves 16-B (bf16) / 32-B (fp32) pieces: 8 consecutive channels of one pixel.
#include "common.h"
#include <type_traits>

namespace {

constexpr int RED_ROWS = 8;  // pixel groups per workgroup in the reduction kernels (256 = 32 x 8)

// ---------------------------------------------------------------------------------------------
// row pre-reduction: part[nrows][ncols] -> its own first R rows (row r <- sum of rows r, r+R, r+2R, ...).
// The finalize kernels below are one workgroup per 32 channels, which leaves a 4096-row partial buffer to
// two or four CUs; this spreads the bulk of the sum over (ncols/32) x R workgroups first.  In place and
// race free: row r is read and written by workgroup r only.
// ---------------------------------------------------------------------------------------------
constexpr int PRE_R = 16;
__global__ __launch_bounds__(1024) void prereduce_rows_kernel(float* __restrict__ part, int nrows, int ncols) {
  __shared__ double s1[32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, r = blockIdx.y;
  double a = 0.0;
  if (c < ncols)
    for (int i = r + PRE_R * rg; i < nrows; i += PRE_R * 32) a += (double)part[(long)i * ncols + c];
  s1[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && c < ncols) {
    for (int k = 1; k < 32; ++k) a += s1[k][cl];
    part[(long)r * ncols + c] = (float)a;
  }
}

// returns the row count the finalize kernel should read
inline int prereduce(float* part, int nrows, int ncols, hipStream_t s) {
  if (nrows <= 4 * PRE_R) return nrows;
  hipLaunchKernelGGL(prereduce_rows_kernel, dim3(cdiv(ncols, 32), PRE_R), dim3(1024), 0, s, part, nrows, ncols);
  return PRE_R;
}

// ---------------------------------------------------------------------------------------------
// statistics finalize.  Partial sums are channel-major, part[2][C][nblk] (which, channel, producer workgroup), so
// one channel's nblk values are contiguous: TPC threads (a wave, or the whole workgroup for long rows) read them
// with 16-byte loads, accumulate in double and fold with shuffles - one launch whatever nblk is.
// ---------------------------------------------------------------------------------------------
template <int TPC>
__device__ __forceinline__ void channel_sums(const float* __restrict__ part, int nblk, int C, int c, int li, double& a,
                                             double& b) {
  const float* __restrict__ p0 = part + (long)c * nblk;
  const float* __restrict__ p1 = part + ((long)C + c) * nblk;
  a = 0.0; b = 0.0;
  if ((nblk & 3) == 0) {
    const float4* __restrict__ q0 = reinterpret_cast<const float4*>(p0);
    const float4* __restrict__ q1 = reinterpret_cast<const float4*>(p1);
    for (int i = li; i < (nblk >> 2); i += TPC) {
      const float4 u = q0[i], v = q1[i];
      a += ((double)u.x + (double)u.y) + ((double)u.z + (double)u.w);
      b += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
  } else {
    for (int i = li; i < nblk; i += TPC) { a += (double)p0[i]; b += (double)p1[i]; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
  if (TPC == 256) {                      // c is workgroup-uniform here: every thread reaches the barrier
    __shared__ double r[2][4];
    if ((threadIdx.x & 63) == 0) { r[0][threadIdx.x >> 6] = a; r[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    a = (r[0][0] + r[0][1]) + (r[0][2] + r[0][3]);
    b = (r[1][0] + r[1][1]) + (r[1][2] + r[1][3]);
  }
}
inline int finalize_tpc(int nblk) { return nblk > 256 ? 256 : 64; }

template <int TPC>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long* num_batches, float momentum,
                                   float eps, float* mean_out, float* invstd_out, float* scale_out,
                                   float* shift_out, const float* __restrict__ conv_bias) {
  const int c = blockIdx.x * (256 / TPC) + threadIdx.x / TPC, li = threadIdx.x % TPC;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
  if (c >= C) return;                    // wave-uniform (TPC >= 64); never taken when TPC == 256 (grid = C)
  double a, b;
  channel_sums<TPC>(part, nblk, C, c, li, a, b);
  if (li == 0) {
    const double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    scale_out[c] = sc;
    shift_out[c] = beta[c] - (float)mean * sc;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
      // conv_bias: the statistics were taken of the convolution WITHOUT its bias (the persistent conv kernel adds none:
      // a bias in front of a train-mode BatchNorm cancels in the normalised output); the running mean is the one of the
      // biased output, as nn.BatchNorm2d behind nn.Conv2d(bias=True) keeps it
      const float mb = conv_bias ? (float)mean + conv_bias[c] : (float)mean;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mb;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// SyncBatchNorm (configs/trainer/ddp.yaml:9): the rank-local per-channel (sum, sum of squares) in exchange form,
// sums[2][C]; the host all-reduces them over the ranks and hands them back to bn_finalize_kernel as a one-block partial
// buffer with the global element count.
template <int TPC>
__global__ __launch_bounds__(256) void bn_partial_sums_kernel(const float* __restrict__ part, int nblk, int C,
                                                              float* __restrict__ sums) {
  const int c = blockIdx.x * (256 / TPC) + threadIdx.x / TPC, li = threadIdx.x % TPC;
  if (c >= C) return;
  double a, b;
  channel_sums<TPC>(part, nblk, C, c, li, a, b);
  if (li == 0) { sums[c] = (float)a; sums[C + c] = (float)b; }
}

__global__ void bn_eval_prepare_kernel(int C, const float* gamma, const float* beta, const float* rmean,
                                       const float* rvar, float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rvar[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rmean[c] * sc;
}

// ---------------------------------------------------------------------------------------------
// forward apply (+ optional fused 2x2 max pool)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_relu_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                     const float* __restrict__ shift, T* __restrict__ y, int ldy,
                                     T* __restrict__ pool, int ldp, int B, int H, int W, int C) {
  const int cp = C >> 3;
  const int Hw = (H + 1) >> 1, Ww = (W + 1) >> 1, Hp = H >> 1, Wp = W >> 1;
  const long total = (long)B * Hw * Ww * cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cp) * 8;
    long t = i / cp;
    const int wx = (int)(t % Ww); t /= Ww;
    const int wy = (int)(t % Hw);
    const int n = (int)(t / Hw);
    float sc[8], sh[8], mx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[c8 + k]; sh[k] = shift[c8 + k]; mx[k] = 0.f; }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int yy = wy * 2 + dy, xx = wx * 2 + dx;
        if (yy < H && xx < W) {
          const long pix = ((long)n * H + yy) * W + xx;
          f32x8 v = load8(x + pix * ldx + c8);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            v.v[k] = fmaxf(fmaf(v.v[k], sc[k], sh[k]), 0.f);   // the backward recomputes exactly this expression
            // pool on the value as stored (bf16-rounded in bf16 mode) so backward can recompute argmax
            const float st = to_f32(from_f32<T>(v.v[k]));
            mx[k] = fmaxf(mx[k], st);
          }
          store8(y + pix * ldy + c8, v);
        }
      }
    if (pool && wy < Hp && wx < Wp) {
      f32x8 m;
#pragma unroll
      for (int k = 0; k < 8; ++k) m.v[k] = mx[k];
      store8(pool + (((long)n * Hp + wy) * Wp + wx) * ldp + c8, m);
    }
  }
}

template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ x, int ldx, T* __restrict__ pool, int ldp, int B, int H,
                                int W, int C) {
  const int cp = C >> 3, Hp = H >> 1, Wp = W >> 1;
  const long total = (long)B * Hp * Wp * cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cp) * 8;
    long t = i / cp;
    const int wx = (int)(t % Wp); t /= Wp;
    const int wy = (int)(t % Hp);
    const int n = (int)(t / Hp);
    f32x8 m;
#pragma unroll
    for (int k = 0; k < 8; ++k) m.v[k] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const long pix = ((long)n * H + wy * 2 + dy) * W + wx * 2 + dx;
        const f32x8 v = load8(x + pix * ldx + c8);
#pragma unroll
        for (int k = 0; k < 8; ++k) m.v[k] = fmaxf(m.v[k], v.v[k]);
      }
    store8(pool + (((long)n * Hp + wy) * Wp + wx) * ldp + c8, m);
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// Gradient reaching the ReLU output at the 4 pixels of one 2x2 window, 8 channels:
//   g = g1 (optional dense / sliced gradient) + max-pool scatter of gp (optional).
// The pool winner is the FIRST maximum in (0,0),(0,1),(1,0),(1,1) order, as ATen's max_pool2d picks
// it; ties at 0 are irrelevant because the ReLU mask kills them.
// The tensors stay in their storage form (16 B per pixel-piece in bf16) until a channel is evaluated, so a thread
// holds 36 registers of window data instead of 96 fp32 ones; the kernels below then have room for 4+ waves/SIMD.
typedef __attribute__((ext_vector_type(8))) float f32x8v;
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> { typedef bf16x8 type; };
template <> struct Vec8<float> { typedef f32x8v type; };

__device__ __forceinline__ bf16x8 vload8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ f32x8v vload8(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void vstore8(bf16_t* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
__device__ __forceinline__ void vstore8(float* p, f32x8v v) {
  *reinterpret_cast<f32x4*>(p) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
  *reinterpret_cast<f32x4*>(p + 4) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
}

template <typename T>
struct WindowGrad {
  typedef typename Vec8<T>::type V;
  V x0, x1, x2, x3, g0, g1v, g2, g3, pr;   // conv output / dense gradient at the 4 pixels, pooled gradient
  int okmask;                               // bit q: pixel q inside the image; bit 4: window has a pool output
  // The ReLU output is not read back: y = relu(x*scale+shift) is recomputed from the saved conv output with the
  // forward's exact fp32 expression (and rounded to the storage type like the forward did), so the mask and the
  // pool winner are bit-identical to the forward's while the backward moves 2 B/element less per pass.
  __device__ __forceinline__ void load(const T* __restrict__ g1, int ldg1, const T* __restrict__ gp, int ldgp,
                                       const T* __restrict__ x, int ldx, int n, int wy, int wx, int H, int W,
                                       int c8) {
    const int Hp = H >> 1, Wp = W >> 1;
    const V zero = {};
    const int yy = wy * 2, xx = wx * 2;
    const bool r1 = yy + 1 < H, c1 = xx + 1 < W;
    okmask = 1 | (c1 ? 2 : 0) | (r1 ? 4 : 0) | ((r1 && c1) ? 8 : 0);
    const long p00 = ((long)n * H + yy) * W + xx;
    x0 = vload8(x + p00 * ldx + c8);
    x1 = c1 ? vload8(x + (p00 + 1) * ldx + c8) : zero;
    x2 = r1 ? vload8(x + (p00 + W) * ldx + c8) : zero;
    x3 = (r1 && c1) ? vload8(x + (p00 + W + 1) * ldx + c8) : zero;
    if (g1) {
      g0 = vload8(g1 + p00 * ldg1 + c8);
      g1v = c1 ? vload8(g1 + (p00 + 1) * ldg1 + c8) : zero;
      g2 = r1 ? vload8(g1 + (p00 + W) * ldg1 + c8) : zero;
      g3 = (r1 && c1) ? vload8(g1 + (p00 + W + 1) * ldg1 + c8) : zero;
    } else {
      g0 = zero; g1v = zero; g2 = zero; g3 = zero;
    }
    const bool pooled = gp && wy < Hp && wx < Wp;
    okmask |= pooled ? 16 : 0;
    pr = pooled ? vload8(gp + (((long)n * Hp + wy) * Wp + wx) * ldgp + c8) : zero;
  }
  __device__ __forceinline__ bool ok(int q) const { return (okmask >> q) & 1; }
  // channel k of the window: dz[q] = g * (y > 0), xh[q] = normalised conv output
  __device__ __forceinline__ void eval(int k, float sc, float sh, float mu, float is, float (&dz)[4],
                                       float (&xh)[4]) const {
    const float b[4] = {(float)x0[k], (float)x1[k], (float)x2[k], (float)x3[k]};
    dz[0] = (float)g0[k]; dz[1] = (float)g1v[k]; dz[2] = (float)g2[k]; dz[3] = (float)g3[k];
    float yv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      yv[q] = ok(q) ? to_f32(from_f32<T>(fmaxf(fmaf(b[q], sc, sh), 0.f))) : 0.f;
      xh[q] = ok(q) ? (b[q] - mu) * is : 0.f;
    }
    if (okmask & 16) {
      int best = 0;
      float bv = yv[0];
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (yv[q] > bv) { bv = yv[q]; best = q; }
      const float p = (float)pr[k];
#pragma unroll
      for (int q = 0; q < 4; ++q) dz[q] += (q == best) ? p : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (!(yv[q] > 0.f)) dz[q] = 0.f;
  }
};

// Thread layout of the two backward kernels: a workgroup covers PCB channel pieces (8 channels each,
// PCB = min(32, pow2 >= C/8)) x WL = 256/PCB window slots, piece index fastest across lanes, so a wave
// reads whole NHWC pixel rows (full cache lines) instead of 32-channel slivers.
__device__ __forceinline__ int pcb_of(int C) {
  int p = 1;
  while (p < 32 && p * 8 < C) p <<= 1;
  return p;
}

// partial[2][C][nblk] (channel-major): sum dz, sum dz*xhat.  grid = (channel groups, nblk)
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const T* g1, int ldg1, const T* gp, int ldgp, const float* scale, const float* shift,
                                          const T* x, int ldx, const float* mean, const float* invstd,
                                          float* part, int B, int H, int W, int C) {
  const int Hw = (H + 1) >> 1, Ww = (W + 1) >> 1;
  const long nwin = (long)B * Hw * Ww;
  const int PCB = pcb_of(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int c8 = (blockIdx.x * PCB + pc) * 8;
  // parity mode accumulates in fp64 (as ATen's CPU batch_norm backward does): sum(dz) cancels heavily once
  // training is under way, and its error is multiplied by the all-positive activations in the next wgrad
  using Acc = typename std::conditional<std::is_same<T, float>::value, double, float>::type;
  Acc s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0; s2[k] = 0; }
  // per-channel constants live in LDS (a thread's 8 channels x 4 arrays would cost 32 VGPRs for the whole loop)
  __shared__ float kc[4][256];
  for (int i = threadIdx.x; i < PCB * 8; i += 256) {
    const int c = blockIdx.x * PCB * 8 + i;
    const bool in = c < C;
    kc[0][i] = in ? scale[c] : 0.f; kc[1][i] = in ? shift[c] : 0.f;
    kc[2][i] = in ? mean[c] : 0.f;  kc[3][i] = in ? invstd[c] : 0.f;
  }
  __syncthreads();
  if (c8 < C) {
    // 32-bit window index (host checks B*Hw*Ww < 2^31): the 64-bit div/mod sequences cost more than the math
    for (int w = blockIdx.y * WL + wl; w < (int)nwin; w += gridDim.y * WL) {
      const int t1 = w / Ww, wx = w - t1 * Ww;
      const int n = t1 / Hw, wy = t1 - n * Hw;
      WindowGrad<T> wg;
      wg.load(g1, ldg1, gp, ldgp, x, ldx, n, wy, wx, H, W, c8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float dz[4], xh[4];
        const int ci = pc * 8 + k;
        wg.eval(k, kc[0][ci], kc[1][ci], kc[2][ci], kc[3][ci], dz, xh);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { a += dz[q]; b += dz[q] * xh[q]; }
        s1[k] += (Acc)a;
        s2[k] += (Acc)b;
        __builtin_amdgcn_sched_barrier(0);    // one channel at a time: interleaving all eight costs 60+ VGPRs
      }
    }
  }
  __shared__ Acc red[2][2304];
  const int rowlen = PCB * 8 + 1;
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][wl * rowlen + pc * 8 + k] = s1[k]; red[1][wl * rowlen + pc * 8 + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * PCB * 8; i += 256) {
    const int which = i / (PCB * 8), cl = i - which * (PCB * 8);
    Acc s = 0;
    for (int r = 0; r < WL; ++r) s += red[which][r * rowlen + cl];
    const int c = blockIdx.x * PCB * 8 + cl;
    if (c < C) part[((long)which * C + c) * gridDim.y + blockIdx.y] = (float)s;
  }
}

// sums partial[2][C][nblk] -> dgamma, dbeta (accumulate optional) and the two per-channel means
template <int TPC>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* part, int nblk, int C, double count, float* dgamma,
                                       float* dbeta, int accumulate, float* c1, float* c2, float* zero_out) {
  const int c = blockIdx.x * (256 / TPC) + threadIdx.x / TPC, li = threadIdx.x % TPC;
  if (c >= C) return;
  double a, b;
  channel_sums<TPC>(part, nblk, C, c, li, a, b);
  if (li == 0) {
    dbeta[c] = accumulate ? dbeta[c] + (float)a : (float)a;
    dgamma[c] = accumulate ? dgamma[c] + (float)b : (float)b;
    c1[c] = (float)(a / count);
    c2[c] = (float)(b / count);
    if (zero_out) zero_out[c] = 0.f;
  }
}

// dx = gamma*invstd*(dz - c1 - xhat*c2); optional partial sums of dx over pixels (conv-bias gradient)
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const T* g1, int ldg1, const T* gp, int ldgp, const float* scale, const float* shift,
                                         const T* x, int ldx, const float* mean, const float* invstd,
                                         const float* gamma, const float* c1, const float* c2, T* dx, int lddx,
                                         float* dxsum_part, int B, int H, int W, int C) {
  const int Hw = (H + 1) >> 1, Ww = (W + 1) >> 1;
  const long nwin = (long)B * Hw * Ww;
  const int PCB = pcb_of(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int c8 = (blockIdx.x * PCB + pc) * 8;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  __shared__ float kc[7][256];
  for (int i = threadIdx.x; i < PCB * 8; i += 256) {
    const int c = blockIdx.x * PCB * 8 + i;
    const bool in = c < C;
    kc[0][i] = in ? scale[c] : 0.f; kc[1][i] = in ? shift[c] : 0.f;
    kc[2][i] = in ? mean[c] : 0.f;  kc[3][i] = in ? invstd[c] : 0.f;
    kc[4][i] = in ? gamma[c] * invstd[c] : 0.f;
    kc[5][i] = in ? c1[c] : 0.f;    kc[6][i] = in ? c2[c] : 0.f;
  }
  __syncthreads();
  if (c8 < C) {
    // 32-bit window index (host checks B*Hw*Ww < 2^31): the 64-bit div/mod sequences cost more than the math
    for (int w = blockIdx.y * WL + wl; w < (int)nwin; w += gridDim.y * WL) {
      const int t1 = w / Ww, wx = w - t1 * Ww;
      const int n = t1 / Hw, wy = t1 - n * Hw;
      WindowGrad<T> wg;
      wg.load(g1, ldg1, gp, ldgp, x, ldx, n, wy, wx, H, W, c8);
      typename Vec8<T>::type o0, o1, o2, o3;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float dz[4], xh[4];
        const int ci = pc * 8 + k;
        wg.eval(k, kc[0][ci], kc[1][ci], kc[2][ci], kc[3][ci], dz, xh);
        const float gak = kc[4][ci], k1k = kc[5][ci], k2k = kc[6][ci];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = gak * (dz[q] - k1k - xh[q] * k2k);
          if (wg.ok(q)) s[k] += v[q];
        }
        o0[k] = from_f32<T>(v[0]); o1[k] = from_f32<T>(v[1]); o2[k] = from_f32<T>(v[2]); o3[k] = from_f32<T>(v[3]);
        __builtin_amdgcn_sched_barrier(0);    // one channel at a time: interleaving all eight costs 60+ VGPRs
      }
      T* const d00 = dx + (((long)n * H + wy * 2) * W + wx * 2) * lddx + c8;
      vstore8(d00, o0);
      if (wg.ok(1)) vstore8(d00 + lddx, o1);
      if (wg.ok(2)) vstore8(d00 + (long)W * lddx, o2);
      if (wg.ok(3)) vstore8(d00 + (long)(W + 1) * lddx, o3);
    }
  }
  if (dxsum_part) {
    __shared__ float red[2304];
    const int rowlen = PCB * 8 + 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wl * rowlen + pc * 8 + k] = s[k];
    __syncthreads();
    for (int cl = threadIdx.x; cl < PCB * 8; cl += 256) {
      float t = 0.f;
      for (int r = 0; r < WL; ++r) t += red[r * rowlen + cl];
      const int c = blockIdx.x * PCB * 8 + cl;
      if (c < C) dxsum_part[(long)blockIdx.y * C + c] = t;
    }
  }
}

// ---- flat variants (no max-pool gradient): one pixel x 8 channels per thread-iteration, pure streaming ----
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_flat_kernel(const T* __restrict__ g1, int ldg1, const float* __restrict__ scale, const float* __restrict__ shift,
                                               const T* __restrict__ x, int ldx, const float* mean, const float* invstd,
                                               float* part, long npix, int C) {
  const int PCB = pcb_of(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int c8 = (blockIdx.x * PCB + pc) * 8;
  using Acc = typename std::conditional<std::is_same<T, float>::value, double, float>::type;
  Acc s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0; s2[k] = 0; }
  if (c8 < C) {
    float mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { mu[k] = mean[c8 + k]; is[k] = invstd[c8 + k]; sc[k] = scale[c8 + k]; sh[k] = shift[c8 + k]; }
    for (long p = (long)blockIdx.y * WL + wl; p < npix; p += (long)gridDim.y * WL) {
      const f32x8 g = load8(g1 + p * ldg1 + c8);
      const f32x8 b = load8(x + p * ldx + c8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dz = fmaf(b.v[k], sc[k], sh[k]) > 0.f ? g.v[k] : 0.f;   // ReLU mask from the conv output
        s1[k] += (Acc)dz;
        s2[k] += (Acc)(dz * ((b.v[k] - mu[k]) * is[k]));
      }
    }
  }
  __shared__ Acc red[2][2304];
  const int rowlen = PCB * 8 + 1;
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][wl * rowlen + pc * 8 + k] = s1[k]; red[1][wl * rowlen + pc * 8 + k] = s2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * PCB * 8; i += 256) {
    const int which = i / (PCB * 8), cl = i - which * (PCB * 8);
    Acc s = 0;
    for (int r = 0; r < WL; ++r) s += red[which][r * rowlen + cl];
    const int c = blockIdx.x * PCB * 8 + cl;
    if (c < C) part[((long)which * C + c) * gridDim.y + blockIdx.y] = (float)s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_flat_kernel(const T* __restrict__ g1, int ldg1, const float* __restrict__ scale, const float* __restrict__ shift,
                                              const T* __restrict__ x, int ldx, const float* mean, const float* invstd,
                                              const float* gamma, const float* c1, const float* c2, T* __restrict__ dx,
                                              int lddx, float* dxsum_part, long npix, int C, int rev) {
  const int PCB = pcb_of(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int c8 = (blockIdx.x * PCB + pc) * 8;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  if (c8 < C) {
    float ga[8], k1[8], k2[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mu[k] = mean[c8 + k]; is[k] = invstd[c8 + k]; sc[k] = scale[c8 + k]; sh[k] = shift[c8 + k];
      ga[k] = gamma[c8 + k] * is[k]; k1[k] = c1[c8 + k]; k2[k] = c2[c8 + k];
    }
    // rev: walk the tensor from its end - the reduction pass just read it front to back, so its tail is what the
    // memory-side cache still holds
    const long p0 = (long)blockIdx.y * WL + wl, pstep = (long)gridDim.y * WL;
    const long nit = p0 < npix ? (npix - p0 + pstep - 1) / pstep : 0;
    for (long it = 0; it < nit; ++it) {
      const long p = p0 + (rev ? nit - 1 - it : it) * pstep;
      const f32x8 g = load8(g1 + p * ldg1 + c8);
      const f32x8 b = load8(x + p * ldx + c8);
      f32x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dz = fmaf(b.v[k], sc[k], sh[k]) > 0.f ? g.v[k] : 0.f;
        o.v[k] = ga[k] * (dz - k1[k] - ((b.v[k] - mu[k]) * is[k]) * k2[k]);
        s[k] += o.v[k];
      }
      store8(dx + p * lddx + c8, o);
    }
  }
  if (dxsum_part) {
    __shared__ float red[2304];
    const int rowlen = PCB * 8 + 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wl * rowlen + pc * 8 + k] = s[k];
    __syncthreads();
    for (int cl = threadIdx.x; cl < PCB * 8; cl += 256) {
      float t = 0.f;
      for (int r = 0; r < WL; ++r) t += red[r * rowlen + cl];
      const int c = blockIdx.x * PCB * 8 + cl;
      if (c < C) dxsum_part[(long)blockIdx.y * C + c] = t;
    }
  }
}

// out[c] (+)= sum_i part[i][c]
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* part, int nblk, int C, float* out, int accumulate) {
  __shared__ double s1[32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0;
  if (c < C)
    for (int i = rg; i < nblk; i += 32) a += (double)part[(long)i * C + c];
  s1[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && c < C) {
    for (int k = 1; k < 32; ++k) a += s1[k][cl];
    out[c] = accumulate ? out[c] + (float)a : (float)a;
  }
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  if (b > 2048 * 4) b = 2048 * 4;
  if (b < 1) b = 1;
  return (int)b;
}

inline int host_pcb(int C) {
  int p = 1;
  while (p < 32 && p * 8 < C) p <<= 1;
  return p;
}

// partial rows per channel group: enough workgroups (~1536 in all) to fill 256 CUs a few times over
inline int red_blocks(int B, int H, int W, int C) {
  const long nwin = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
  const int pcb = host_pcb(C), wl = 256 / pcb;
  long want = 1536 / cdiv(C / 8, pcb);
  if (want < 1) want = 1;
  long nb = (nwin + wl - 1) / wl;
  if (nb > want) nb = want;
  if (nb < 1) nb = 1;
  return (int)nb;
}

}  // namespace

extern "C" int s2s_bn_finalize_b(const float* part, int nblk, int C, long count, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, long* num_batches,
                                 float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                                 const float* conv_bias, void* stream) {
  if (!part || !gamma || !beta || !mean || !invstd || !scale || !shift) return S2S_ERR_NULL;
  if (nblk <= 0 || C <= 0 || count <= 0) return S2S_ERR_SHAPE;
  if (finalize_tpc(nblk) == 256)
    hipLaunchKernelGGL(bn_finalize_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, part, nblk, C,
                       (double)count, gamma, beta, running_mean, running_var, num_batches, momentum, eps, mean,
                       invstd, scale, shift, conv_bias);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, part, nblk, C,
                       (double)count, gamma, beta, running_mean, running_var, num_batches, momentum, eps, mean,
                       invstd, scale, shift, conv_bias);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_bn_finalize(const float* part, int nblk, int C, long count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, long* num_batches,
                               float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                               void* stream) {
  return s2s_bn_finalize_b(part, nblk, C, count, gamma, beta, running_mean, running_var, num_batches, momentum, eps, mean,
                           invstd, scale, shift, nullptr, stream);
}

extern "C" int s2s_bn_partial_sums(const float* part, int nblk, int C, float* sums, void* stream) {
  if (!part || !sums) return S2S_ERR_NULL;
  if (nblk <= 0 || C <= 0) return S2S_ERR_SHAPE;
  if (finalize_tpc(nblk) == 256)
    hipLaunchKernelGGL(bn_partial_sums_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, part, nblk, C, sums);
  else
    hipLaunchKernelGGL(bn_partial_sums_kernel<64>, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, part, nblk, C, sums);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_bn_eval_prepare(int C, const float* gamma, const float* beta, const float* rmean,
                                   const float* rvar, float eps, float* scale, float* shift, void* stream) {
  if (!gamma || !beta || !rmean || !rvar || !scale || !shift) return S2S_ERR_NULL;
  if (C <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                     rmean, rvar, eps, scale, shift);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_bn_relu_apply(int dtype, const void* x, int ldx, const float* scale, const float* shift, void* y,
                                 int ldy, void* pool, int ldp, int B, int H, int W, int C, void* stream) {
  if (!x || !scale || !shift || !y) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (ldy % 8) || (pool && (ldp % 8)))
    return S2S_ERR_SHAPE;
  const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(bn_relu_apply_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, (const bf16_t*)x, ldx,
                       scale, shift, (bf16_t*)y, ldy, (bf16_t*)pool, ldp, B, H, W, C);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(bn_relu_apply_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x, ldx,
                       scale, shift, (float*)y, ldy, (float*)pool, ldp, B, H, W, C);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_maxpool2(int dtype, const void* x, int ldx, void* pool, int ldp, int B, int H, int W, int C,
                            void* stream) {
  if (!x || !pool) return S2S_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2 || C <= 0 || (C % 8) || (ldx % 8) || (ldp % 8)) return S2S_ERR_SHAPE;
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(maxpool2_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, (const bf16_t*)x, ldx,
                       (bf16_t*)pool, ldp, B, H, W, C);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x, ldx,
                       (float*)pool, ldp, B, H, W, C);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// number of partial rows s2s_bn_relu_bwd needs in each of its two workspaces
extern "C" int s2s_bn_bwd_blocks(int B, int H, int W, int C) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return S2S_ERR_SHAPE;
  return red_blocks(B, H, W, C);
}

// Whole BN+ReLU(+pool scatter) backward: reduce -> finalize -> apply (-> conv-bias gradient).
//   g1/gp : gradient wrt the ReLU output (dense or channel slice) / wrt the pooled output (may be null)
//   scale/shift: the forward's folded BN affine (y = relu(x*scale+shift) is recomputed, not read); x: saved conv output;
//   work: float[4*blocks*C + 2*C]
// phases: 1 = reduction pass + per-channel finalize (dgamma, dbeta from the LOCAL sums; the two means c1, c2 = local
// sums / count_total into work[4 nb C .. +2C)), 2 = the apply pass reading c1, c2 from there, 3 = both.  SyncBatchNorm
// runs phase 1 with the GLOBAL element count, all-reduces the 2C floats over the ranks and runs phase 2: the same
// exchange as torch.nn.SyncBatchNorm's backward (sum_dy, sum_dy_xmu), weight / bias gradients stay rank-local sums.
extern "C" int s2s_bn_relu_bwd_phase(int dtype, const void* g1, int ldg1, const void* gp, int ldgp, const float* scale,
                               const float* shift, const void* x, int ldx, const float* mean, const float* invstd, const float* gamma,
                               float* dgamma, float* dbeta, float* dbias_conv, int accumulate, void* dx, int lddx,
                               float* work, int B, int H, int W, int C, long count_total, int phases, void* stream) {
  if (phases < 1 || phases > 3 || count_total < (long)B * H * W) return S2S_ERR_SHAPE;
  if ((!g1 && !gp) || !scale || !shift || !x || !mean || !invstd || !gamma || !dgamma || !dbeta || !dx || !work) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (lddx % 8) ||
      (g1 && (ldg1 % 8)) || (gp && (ldgp % 8)))
    return S2S_ERR_SHAPE;
  if ((long)B * ((H + 1) / 2) * ((W + 1) / 2) >= (1L << 31) - (1 << 20)) return S2S_ERR_SHAPE;   // 32-bit window index
  const int nb = red_blocks(B, H, W, C);
  float* part = work;                       // [2][C][nb]
  float* part2 = work + (long)nb * 2 * C;   // [nb][C]
  float* c1 = part2 + (long)nb * 2 * C;     // [C]
  float* c2 = c1 + C;
  const double count = (double)count_total;
  hipStream_t s = (hipStream_t)stream;
  // Gradient of the conv bias that feeds a train-mode BatchNorm: sum over pixels of dx, which the BN backward
  // formula makes identically zero (sum dz - N c1 - c2 sum xhat, with c1 = sum dz / N and sum xhat = 0).  The
  // reference's autograd sums it anyway and gets rounding noise of either sign (|.| ~ 1e-9); by default we write the
  // exact value and skip the partial sums and their two reduction launches per layer.
  float* const dbias_zero = (dbias_conv && !accumulate) ? dbias_conv : nullptr;
  dbias_conv = nullptr;
  dim3 grid(cdiv(C / 8, host_pcb(C)), nb);
  constexpr int bn_rev = 1;      // the flat apply pass walks the tensor back to front: it starts on what the reduction read last
#define S2S_BN_BWD(TT)                                                                                             \
  if (!(phases & 1)) {                                                                                             \
  } else if (gp) {                                                                                                 \
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<TT>, grid, dim3(256), 0, s, (const TT*)g1, ldg1, (const TT*)gp,   \
                       ldgp, scale, shift, (const TT*)x, ldx, mean, invstd, part, B, H, W, C);                \
  } else {                                                                                                         \
    hipLaunchKernelGGL(bn_relu_bwd_reduce_flat_kernel<TT>, grid, dim3(256), 0, s, (const TT*)g1, ldg1,             \
                       scale, shift, (const TT*)x, ldx, mean, invstd, part, (long)B * H * W, C);              \
  }                                                                                                                \
  if (!(phases & 1)) {                                                                                             \
  } else if (finalize_tpc(nb) == 256)                                                                              \
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<256>, dim3(C), dim3(256), 0, s, part, nb, C, count, dgamma, dbeta,   \
                       accumulate, c1, c2, dbias_zero);                                                            \
  else                                                                                                             \
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<64>, dim3(cdiv(C, 4)), dim3(256), 0, s, part, nb, C, count, dgamma,  \
                       dbeta, accumulate, c1, c2, dbias_zero);                                                     \
  if (!(phases & 2)) {                                                                                             \
  } else if (gp) {                                                                                                 \
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<TT>, grid, dim3(256), 0, s, (const TT*)g1, ldg1, (const TT*)gp,    \
                       ldgp, scale, shift, (const TT*)x, ldx, mean, invstd, gamma, c1, c2, (TT*)dx, lddx,     \
                       dbias_conv ? part2 : nullptr, B, H, W, C);                                                  \
  } else {                                                                                                         \
    hipLaunchKernelGGL(bn_relu_bwd_apply_flat_kernel<TT>, grid, dim3(256), 0, s, (const TT*)g1, ldg1,              \
                       scale, shift, (const TT*)x, ldx, mean, invstd, gamma, c1, c2, (TT*)dx, lddx,           \
                       dbias_conv ? part2 : nullptr, (long)B * H * W, C, bn_rev);                                  \
  }
  if (dtype == S2S_BF16) { S2S_BN_BWD(bf16_t) }
  else if (dtype == S2S_F32) { S2S_BN_BWD(float) }
  else return S2S_ERR_DTYPE;
#undef S2S_BN_BWD
  if (dbias_conv && (phases & 2))
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, 32)), dim3(1024), 0, s, part2, prereduce(part2, nb, C, s),
                       C, dbias_conv, accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_bn_relu_bwd(int dtype, const void* g1, int ldg1, const void* gp, int ldgp, const float* scale,
                               const float* shift, const void* x, int ldx, const float* mean, const float* invstd, const float* gamma,
                               float* dgamma, float* dbeta, float* dbias_conv, int accumulate, void* dx, int lddx,
                               float* work, int B, int H, int W, int C, void* stream) {
  return s2s_bn_relu_bwd_phase(dtype, g1, ldg1, gp, ldgp, scale, shift, x, ldx, mean, invstd, gamma, dgamma, dbeta,
                               dbias_conv, accumulate, dx, lddx, work, B, H, W, C, (long)B * H * W, 3, stream);
}
