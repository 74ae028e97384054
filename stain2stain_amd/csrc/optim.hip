// Optimiser step, weight packing and boundary layout conversion.
//
//   s2s_adam_step       torch.optim.Adam (coupled L2 weight decay, no amsgrad) over one flat fp32
//                       parameter buffer -- the optimiser the reference builds in configure_optimizers
//                       (src/models/conditional_flow_matching.py:112-131, configs/model/*.yaml:3-7).
//   s2s_pack_conv3x3    OIHW fp32 master weights -> the two MFMA operand layouts of conv3x3_mfma.hip:
//                       forward  Wf[cin/32][tap][cout][32]  and data-gradient
//                       Wd[cout/32][tap'][cin][32] = W[co][ci][2-kh'][2-kw'] (flipped, transposed); slabs in K-loop order.
//   s2s_nchw_to_nhwc / s2s_nhwc_to_nchw   fp32 NCHW <-> NHWC(T) at the module boundary (the reference's
//                       boundary layout is NCHW contiguous fp32).
#include "common.h"

namespace {

__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                          float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                                          float weight_decay, float bc1, float bc2_sqrt, float grad_scale) {
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    float gi = g[i] * grad_scale;
    if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float bc1, float bc2_sqrt, float grad_scale) {
  adam_body(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
}

// The same step with its scalars read from DEVICE memory: hyper[8] = {lr, beta1, beta2, eps, weight_decay, bc1,
// sqrt(bc2), grad_scale}.  A hipGraph-captured training step replays this launch unchanged while the host refreshes the
// eight floats (step count -> bias corrections, scheduler -> lr) before every replay.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long n, const float* __restrict__ hyper) {
  adam_body(p, g, m, v, n, hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7]);
}

// The same step over a LIST of tensors in one launch (the drop-in path: ordinary module parameters with the gradients
// autograd left in ``.grad``, stain2stain_amd.FusedAdam): desc[t] = {p, g, m, v, n, first block}; a block owns 4096
// consecutive elements of one tensor.  Element for element the arithmetic of adam_body.
struct AdamDesc {
  long p, g, m, v, n, start;
};
constexpr int ADAM_MULTI_CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamDesc* __restrict__ desc, int nt, float lr, float beta1,
                                                         float beta2, float eps, float weight_decay, float bc1,
                                                         float bc2_sqrt, float grad_scale) {
  int lo = 0, hi = nt - 1;                       // last tensor whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].start <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const AdamDesc d = desc[lo];
  float* __restrict__ p = reinterpret_cast<float*>(d.p);
  const float* __restrict__ g = reinterpret_cast<const float*>(d.g);
  float* __restrict__ m = reinterpret_cast<float*>(d.m);
  float* __restrict__ v = reinterpret_cast<float*>(d.v);
  const long base = ((long)blockIdx.x - d.start) * ADAM_MULTI_CHUNK;
  const float step_size = lr / bc1;
#pragma unroll 4
  for (int k = 0; k < ADAM_MULTI_CHUNK / 256; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i < d.n) {
      const float pi = p[i];
      float gi = g[i] * grad_scale;
      if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
      const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
      const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      p[i] = pi - step_size * (mi / denom);
      m[i] = mi;
      v[i] = vi;
    }
  }
}

template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout,
                                    int Cin) {
  const int nci = (Cin + 31) / 32, nco = (Cout + 31) / 32;
  const long nf = 9L * nci * Cout * 32, nd = 9L * nco * Cin * 32;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (long)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int k = (int)(i & 31);
      long t = i >> 5;
      const int co = (int)(t % Cout); t /= Cout;
      const int tap = (int)(t % 9);
      const int c = (int)(t / 9);
      const int ci = c * 32 + k;
      wf[i] = from_f32<T>(ci < Cin ? w[((long)co * Cin + ci) * 9 + tap] : 0.f);
    } else if (wd) {
      const long j = i - nf;
      const int k = (int)(j & 31);
      long t = j >> 5;
      const int ci = (int)(t % Cin); t /= Cin;
      const int tap = (int)(t % 9);
      const int c = (int)(t / 9);
      const int co = c * 32 + k;
      wd[j] = from_f32<T>(co < Cout ? w[((long)co * Cin + ci) * 9 + (8 - tap)] : 0.f);
    }
  }
}

// All conv layers in one launch, LDS-tiled so that both the OIHW reads and the two packed writes are coalesced.
// desc[l] = {w_oihw ptr, w_fwd ptr, w_dgrad ptr, Cout, Cin, first tile} (6 x int64 per layer); one workgroup
// repacks a 32(co) x 32(ci) x 9(tap) tile: rows of 32*9 contiguous floats in, 2-KB [32][32] bf16/fp32 blocks out.
struct PackDesc {
  long w, wf, wd, cout, cin, start;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_conv3x3_batched_kernel(const PackDesc* __restrict__ desc, int nlayers) {
  __shared__ float tile[32][289];
  int l = 0;
  while (l + 1 < nlayers && desc[l + 1].start <= (long)blockIdx.x) ++l;
  const PackDesc d = desc[l];
  const float* __restrict__ w = reinterpret_cast<const float*>(d.w);
  T* __restrict__ wf = reinterpret_cast<T*>(d.wf);
  T* __restrict__ wd = reinterpret_cast<T*>(d.wd);
  const int Cout = (int)d.cout, Cin = (int)d.cin;
  const int nci = (Cin + 31) / 32;
  const int t = blockIdx.x - (int)d.start;
  const int co0 = (t / nci) * 32, ci0 = (t % nci) * 32;
  // rows of 288 contiguous floats (32 ci x 9 taps): 16-byte loads when the tile is whole and the row base aligned
  const bool whole = co0 + 32 <= Cout && ci0 + 32 <= Cin && ((Cin * 9) % 4) == 0;
  if (whole) {
    for (int idx = threadIdx.x; idx < 32 * 72; idx += 256) {
      const int co = idx / 72, j4 = idx - co * 72;
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + ((long)(co0 + co) * Cin + ci0) * 9 + j4 * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) tile[co][j4 * 4 + i] = v[i];
    }
  } else {
    for (int idx = threadIdx.x; idx < 32 * 288; idx += 256) {
      const int co = idx / 288, j = idx - co * 288;
      const int ci = ci0 + j / 9;
      tile[co][j] = (co0 + co < Cout && ci < Cin) ? w[((long)(co0 + co) * Cin + ci0) * 9 + j] : 0.f;
    }
  }
  __syncthreads();
  const int c = ci0 / 32, cp = co0 / 32;
  // 8 consecutive k of one (tap, row) per thread: one 16-byte (bf16) store instead of eight 2-byte ones
  for (int idx = threadIdx.x; idx < 9 * 128; idx += 256) {
    const int k8 = (idx & 3) * 8, row = (idx >> 2) & 31, tap = idx >> 7;
    // forward: Wf[c][tap][co0+row][k = ci]
    if (co0 + row < Cout) {
      T* dst = wf + (((long)c * 9 + tap) * Cout + co0 + row) * 32 + k8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = from_f32<T>(tile[row][(k8 + i) * 9 + tap]);
    }
    // data gradient: Wd[cp][tap][ci0+row][k = co] = W[co][ci][8 - tap]
    if (wd && ci0 + row < Cin) {
      T* dst = wd + (((long)cp * 9 + tap) * Cin + ci0 + row) * 32 + k8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = from_f32<T>(tile[k8 + i][row * 9 + (8 - tap)]);
    }
  }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int ldy, int B, int C, int HW) {
  const int cp = C >> 3;
  const long total = (long)B * cp * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % HW);
    long t = i / HW;
    const int c8 = (int)(t % cp) * 8;
    const int n = (int)(t / cp);
    f32x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v.v[k] = x[((long)n * C + c8 + k) * HW + q];
    store8(y + ((long)n * HW + q) * ldy + c8, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y, int B, int C, int HW,
                                    int accumulate) {
  const int cp = C >> 3;
  const long total = (long)B * cp * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % HW);
    long t = i / HW;
    const int c8 = (int)(t % cp) * 8;
    const int n = (int)(t / cp);
    const f32x8 v = load8(x + ((long)n * HW + q) * ldx + c8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float* d = y + ((long)n * C + c8 + k) * HW + q;
      *d = accumulate ? *d + v.v[k] : v.v[k];
    }
  }
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int s2s_adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  if (!p || !g || !m || !v) return S2S_ERR_NULL;
  if (n <= 0 || step <= 0) return S2S_ERR_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                     eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// blocks one tensor of n elements occupies in s2s_adam_multi (the caller accumulates them into desc[t][5])
extern "C" long s2s_adam_multi_blocks(long n) { return n <= 0 ? S2S_ERR_SHAPE : (n + ADAM_MULTI_CHUNK - 1) / ADAM_MULTI_CHUNK; }

// desc: device long[ntensors][6] = {p, g, m, v (fp32 device pointers), n, first block}; total = all blocks
extern "C" int s2s_adam_multi(const void* desc, int ntensors, long total, int step, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float grad_scale, void* stream) {
  if (!desc) return S2S_ERR_NULL;
  if (ntensors <= 0 || total <= 0 || total >= (1L << 31) || step <= 0) return S2S_ERR_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const AdamDesc*>(desc), ntensors, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2), grad_scale);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, void* stream) {
  if (!p || !g || !m || !v || !hyper) return S2S_ERR_NULL;
  if (n <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// element counts of the two packed layouts (for the caller's allocation)
extern "C" long s2s_pack_conv3x3_fwd_elems(int Cout, int Cin) { return 9L * ((Cin + 31) / 32) * Cout * 32; }
extern "C" long s2s_pack_conv3x3_dgrad_elems(int Cout, int Cin) { return 9L * ((Cout + 31) / 32) * Cin * 32; }

extern "C" int s2s_pack_conv3x3(int dtype, const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin,
                                void* stream) {
  if (!w_oihw || !w_fwd) return S2S_ERR_NULL;
  if (Cout <= 0 || Cin <= 0) return S2S_ERR_SHAPE;
  const long total = s2s_pack_conv3x3_fwd_elems(Cout, Cin) + (w_dgrad ? s2s_pack_conv3x3_dgrad_elems(Cout, Cin) : 0);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(pack_conv3x3_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, w_oihw, (bf16_t*)w_fwd,
                       (bf16_t*)w_dgrad, Cout, Cin);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, w_oihw, (float*)w_fwd,
                       (float*)w_dgrad, Cout, Cin);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// desc: device array of nlayers x 6 int64 {w_oihw, w_fwd, w_dgrad, Cout, Cin, start}; `start` = running sum of
// ceil(Cout/32)*ceil(Cin/32) (32x32 tiles) over the preceding layers, total = that sum over all layers.
extern "C" int s2s_pack_conv3x3_batched(int dtype, const void* desc, int nlayers, long total, void* stream) {
  if (!desc) return S2S_ERR_NULL;
  if (nlayers <= 0 || nlayers > 256 || total <= 0 || total > 0x7fffffffL) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(pack_conv3x3_batched_kernel<bf16_t>, dim3((unsigned)total), dim3(256), 0, s,
                       (const PackDesc*)desc, nlayers);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(pack_conv3x3_batched_kernel<float>, dim3((unsigned)total), dim3(256), 0, s,
                       (const PackDesc*)desc, nlayers);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_nchw_to_nhwc(int dtype, const float* x_nchw, void* y, int ldy, int B, int C, int H, int W,
                                void* stream) {
  if (!x_nchw || !y) return S2S_ERR_NULL;
  if (B <= 0 || C <= 0 || (C % 8) || (ldy % 8) || H <= 0 || W <= 0) return S2S_ERR_SHAPE;
  const long total = (long)B * (C / 8) * H * W;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, x_nchw, (bf16_t*)y, ldy, B, C,
                       H * W);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, x_nchw, (float*)y, ldy, B, C,
                       H * W);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_nhwc_to_nchw(int dtype, const void* x, int ldx, float* y_nchw, int accumulate, int B, int C, int H,
                                int W, void* stream) {
  if (!x || !y_nchw) return S2S_ERR_NULL;
  if (B <= 0 || C <= 0 || (C % 8) || (ldx % 8) || H <= 0 || W <= 0) return S2S_ERR_SHAPE;
  const long total = (long)B * (C / 8) * H * W;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, (const bf16_t*)x, ldx, y_nchw,
                       B, C, H * W, accumulate);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x, ldx, y_nchw, B,
                       C, H * W, accumulate);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
