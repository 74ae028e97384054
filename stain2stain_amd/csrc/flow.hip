// Flow-matching glue around the U-Net: time conditioning path, probability-path sampling, loss.
//
//   s2s_time_embedding   TimeEmbedding.forward (src/models/components/shared_encoder.py:114-135)
//   s2s_linear_*, s2s_silu_*   FlowMatchingDecoder.time_mlp / time_proj (task_decoders.py:82-89,119-121)
//   s2s_cfm_sample       ConditionalFlowMatcher.sample_location_and_conditional_flow with explicit t
//                        (torchcfm 1.0.7; call site src/models/conditional_flow_matching.py:66; sigma term
//                        takes caller-provided noise)
//   s2s_mse_loss         loss = mean((v-u)^2) and dv = 2(v-u)*grad_scale/N (conditional_flow_matching.py:72)
//   s2s_axpy             x += a*y on NCHW fp32 images (the Euler update of the sampler)
// All of these are tiny or purely HBM-bound fp32 kernels.
#include "common.h"

namespace {

__global__ void time_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float step = logf(10000.0f) / (float)(half - 1);
  const float ang = t[b] * expf((float)k * -step);
  out[(long)b * dim + k] = sinf(ang);
  out[(long)b * dim + half + k] = cosf(ang);
}

// y[b][n] = sum_k x[b][k] W[n][k] + bias[n]; one wave per output element
__global__ void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ y, int B, int K, int N) {
  const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= B * N) return;
  const int b = o / N, n = o - b * N;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(x[(long)b * K + k], w[(long)n * K + k], s);
  s = wave_sum(s);
  if (lane == 0) y[o] = s + (bias ? bias[n] : 0.f);
}

// dx[b][k] = sum_n dy[b][n] W[n][k]; workgroup = (b, 64 k-columns), 16 waves split n, lanes walk k (coalesced);
// four independent partial sums per lane keep several loads in flight (the loop is latency-bound otherwise)
__global__ __launch_bounds__(1024) void linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                    float* __restrict__ dx, int B, int K, int N) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y, k = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (k < K) {
    int n = wv;
    for (; n + 48 < N; n += 64) {
      s0 = fmaf(dy[(long)b * N + n], w[(long)n * K + k], s0);
      s1 = fmaf(dy[(long)b * N + n + 16], w[(long)(n + 16) * K + k], s1);
      s2 = fmaf(dy[(long)b * N + n + 32], w[(long)(n + 32) * K + k], s2);
      s3 = fmaf(dy[(long)b * N + n + 48], w[(long)(n + 48) * K + k], s3);
    }
    for (; n < N; n += 16) s0 = fmaf(dy[(long)b * N + n], w[(long)n * K + k], s0);
  }
  red[wv][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wv == 0 && k < K) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][lane];
    dx[(long)b * K + k] = s;
  }
}

// dW[n][k] (+)= sum_b dy[b][n] x[b][k];  db[n] (+)= sum_b dy[b][n]
__global__ void linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    float* __restrict__ dw, float* __restrict__ db, int B, int K, int N,
                                    int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * (K + 1)) return;
  const int n = i / (K + 1), k = i - n * (K + 1);
  float s = 0.f;
  if (k < K) {
    for (int b = 0; b < B; ++b) s = fmaf(dy[(long)b * N + n], x[(long)b * K + k], s);
    dw[(long)n * K + k] = accumulate ? dw[(long)n * K + k] + s : s;
  } else if (db) {
    for (int b = 0; b < B; ++b) s += dy[(long)b * N + n];
    db[n] = accumulate ? db[n] + s : s;
  }
}

__global__ void silu_fwd_kernel(const float* __restrict__ h, float* __restrict__ a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = h[i];
  a[i] = v / (1.f + expf(-v));
}

__global__ void silu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ da, float* __restrict__ dh,
                                int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = h[i];
  const float sg = 1.f / (1.f + expf(-v));
  dh[i] = da[i] * sg * (1.f + v * (1.f - sg));
}

__global__ void cfm_sample_kernel(const f32x4* __restrict__ x0, const f32x4* __restrict__ x1,
                                  const float* __restrict__ t, const f32x4* __restrict__ eps, float sigma,
                                  f32x4* __restrict__ xt, f32x4* __restrict__ ut, long per_sample4, long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const float tb = t[i / per_sample4];
    const f32x4 a = x0[i], b = x1[i];
    f32x4 m = tb * b + (1.f - tb) * a;
    if (eps) m += sigma * eps[i];
    xt[i] = m;
    ut[i] = b - a;
  }
}

__global__ void mse_partial_kernel(const f32x4* __restrict__ v, const f32x4* __restrict__ u,
                                   f32x4* __restrict__ dv, float coef, long total4, double* __restrict__ part) {
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = v[i] - u[i];
    s += (double)(d[0] * d[0] + d[1] * d[1]) + (double)(d[2] * d[2] + d[3] * d[3]);
    if (dv) dv[i] = d * coef;
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void mse_finalize_kernel(const double* part, int n, double inv_count, float* loss) {
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

// Scaled error norm of an embedded Runge-Kutta step (the step-size control of the adaptive sampler):
//   part[blk] = sum over elements of (e / (atol + rtol * max(|y0|, |y1|)))^2
__global__ __launch_bounds__(256) void ode_err_partial_kernel(const float* __restrict__ e, const float* __restrict__ y0,
                                                              const float* __restrict__ y1, float atol, float rtol,
                                                              long n, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float sc = atol + rtol * fmaxf(fabsf(y0[i]), fabsf(y1[i]));
    const float r = e[i] / sc;
    s += (double)r * (double)r;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ode_err_finalize_kernel(const double* part, int nblk, double inv_n, float* out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)sqrt(red[0] * inv_n);
}

__global__ void axpy_kernel(f32x4* __restrict__ x, const f32x4* __restrict__ y, float a, long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x)
    x[i] += a * y[i];
}

__global__ void fill_kernel(float* x, float v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
constexpr int MSE_BLOCKS = 1024;

}  // namespace

extern "C" int s2s_time_embedding(const float* t, float* out, int B, int dim, void* stream) {
  if (!t || !out) return S2S_ERR_NULL;
  if (B <= 0 || dim < 4 || (dim & 1)) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(time_embedding_kernel, dim3(cdiv(B * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, out,
                     B, dim);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                              void* stream) {
  if (!x || !w || !y) return S2S_ERR_NULL;
  if (B <= 0 || K <= 0 || N <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(B * N, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, K, N);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                              int accumulate, int B, int K, int N, void* stream) {
  if (!dy || !x || !w || !dw) return S2S_ERR_NULL;
  if (B <= 0 || K <= 0 || N <= 0) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dx) hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(cdiv(K, 64), B), dim3(1024), 0, s, dy, w, dx, B, K, N);
  hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(cdiv(N * (K + 1), 256)), dim3(256), 0, s, dy, x, dw, db, B, K, N,
                     accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_silu_fwd(const float* h, float* a, int n, void* stream) {
  if (!h || !a) return S2S_ERR_NULL;
  if (n <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, h, a, n);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_silu_bwd(const float* h, const float* da, float* dh, int n, void* stream) {
  if (!h || !da || !dh) return S2S_ERR_NULL;
  if (n <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, h, da, dh, n);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_cfm_sample(const float* x0, const float* x1, const float* t, const float* eps, float sigma,
                              float* xt, float* ut, int B, long per_sample, void* stream) {
  if (!x0 || !x1 || !t || !xt || !ut) return S2S_ERR_NULL;
  if (B <= 0 || per_sample <= 0 || (per_sample % 4)) return S2S_ERR_SHAPE;
  if (((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)xt | (uintptr_t)ut | (uintptr_t)eps) & 15) return S2S_ERR_ALIGN;
  const long total4 = (long)B * per_sample / 4;
  hipLaunchKernelGGL(cfm_sample_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x0,
                     (const f32x4*)x1, t, (sigma != 0.f) ? (const f32x4*)eps : nullptr, sigma, (f32x4*)xt, (f32x4*)ut,
                     per_sample / 4, total4);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// work: double[1024].  loss <- mean((v-u)^2);  dv (optional) <- 2*(v-u)*grad_scale/count
extern "C" int s2s_mse_loss(const float* v, const float* u, float* dv, float grad_scale, float* loss, double* work,
                            long count, void* stream) {
  if (!v || !u || !loss || !work) return S2S_ERR_NULL;
  if (count <= 0 || (count % 4)) return S2S_ERR_SHAPE;
  if (((uintptr_t)v | (uintptr_t)u | (uintptr_t)dv) & 15) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  int nb = ew_grid(count / 4);
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, s, (const f32x4*)v, (const f32x4*)u, (f32x4*)dv,
                     (float)(2.0 * (double)grad_scale / (double)count), count / 4, work);
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, s, work, nb, 1.0 / (double)count, loss);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// out[0] = sqrt(mean((e / (atol + rtol * max(|y0|, |y1|)))^2)); work: double[1024]
extern "C" int s2s_ode_error_norm(const float* e, const float* y0, const float* y1, float atol, float rtol,
                                  float* out, double* work, long n, void* stream) {
  if (!e || !y0 || !y1 || !out || !work) return S2S_ERR_NULL;
  if (n <= 0) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  int nb = ew_grid(n);
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  hipLaunchKernelGGL(ode_err_partial_kernel, dim3(nb), dim3(256), 0, s, e, y0, y1, atol, rtol, n, work);
  hipLaunchKernelGGL(ode_err_finalize_kernel, dim3(1), dim3(256), 0, s, work, nb, 1.0 / (double)n, out);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_axpy(float* x, const float* y, float a, long n, void* stream) {
  if (!x || !y) return S2S_ERR_NULL;
  if (n <= 0 || (n % 4)) return S2S_ERR_SHAPE;
  if (((uintptr_t)x | (uintptr_t)y) & 15) return S2S_ERR_ALIGN;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (f32x4*)x, (const f32x4*)y,
                     a, n / 4);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// t[0..n) <- table[*counter]; *counter += 1.  The time input of a graph-captured Euler step: the node times t_k =
// k / num_steps are tabulated on the host exactly as the eager loop computes them, and the captured kernel walks the table.
__global__ void euler_tick_kernel(float* __restrict__ t, int n, const float* __restrict__ table, int* __restrict__ counter) {
  const int k = *counter;
  const float v = table[k];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) t[i] = v;
  if (threadIdx.x == 0) *counter = k + 1;
}

extern "C" int s2s_euler_tick(float* t, int n, const float* table, int* counter, void* stream) {
  if (!t || !table || !counter) return S2S_ERR_NULL;
  if (n <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(euler_tick_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, n, table, counter);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_fill_f32(float* x, float v, long n, void* stream) {
  if (!x) return S2S_ERR_NULL;
  if (n <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, v, n);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
