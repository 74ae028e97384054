// Loss variants and class conditioning around the same network call.
//
//   ROI-weighted flow-matching MSE     src/models/conditional_flow_matching_masked.py:76-90
//       w = 1 + lam * mask (broadcast over channels);  loss = sum(w (v-u)^2) / (sum(w) + 1e-8)
//   ROI Charbonnier term               src/models/conditional_flow_matching_ROI_loss.py:78-95
//       charb = sqrt((a-b)^2 + eps^2);  roi = sum(charb * m) / (sum(m) * C + 1e-8)      (a = xt, b = x1: data only,
//       no gradient reaches the network -- the reference adds it to the loss value all the same)
//   class conditioning                 src/models/class_conditional_flow_matching.py:39-71 calls net(t, x, y=y) on a
//       third-party U-Net; here (build-defined) a learned table row is added to the sinusoidal time
//       embedding:  e[b,:] = temb[b,:] + table[y[b],:]
#include "common.h"

namespace {

constexpr int LV_BLOCKS = 512;

// MODE 0: s0 = sum w (v-u)^2, s1 = sum w          MODE 1: s0 = sum sqrt((a-b)^2+eps^2) m, s1 = sum m (per pixel)
template <int MODE>
__global__ __launch_bounds__(256) void lv_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ mask, long B, int C, long HW,
                                                        float p0, double* __restrict__ part) {
  double s0 = 0.0, s1 = 0.0;
  const long npix = B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long n = i / HW, q = i - n * HW;
    const float m = mask[i];
    const float w = MODE == 0 ? 1.f + p0 * m : m;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
      const long idx = (n * C + c) * HW + q;
      const float d = a[idx] - b[idx];
      acc += MODE == 0 ? d * d : sqrtf(d * d + p0 * p0);
    }
    s0 += (double)(w * acc);
    s1 += (double)(MODE == 0 ? w * (float)C : m);
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) part[(long)blockIdx.x * 2 + threadIdx.x] = red[threadIdx.x][0];
}

// out[0] = s0 / (s1 * mul + eps_den); sums[0..1] kept for the gradient pass
__global__ void lv_finalize_kernel(const double* part, int nblk, double mul, double eps_den, float* out, double* sums) {
  if (threadIdx.x < 2) {
    double acc = 0.0;
    for (int i = 0; i < nblk; ++i) acc += part[(long)i * 2 + threadIdx.x];
    sums[threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(sums[0] / (sums[1] * mul + eps_den));
}

__global__ __launch_bounds__(256) void wmse_bwd_kernel(const float* __restrict__ v, const float* __restrict__ u,
                                                       const float* __restrict__ mask, long B, int C, long HW,
                                                       float lam, const double* __restrict__ sums, float scale,
                                                       float* __restrict__ dv) {
  const float k = (float)(2.0 * (double)scale / (sums[1] + 1e-8));
  const long total = B * C * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / (C * HW), q = i % HW;
    dv[i] = k * (1.f + lam * mask[n * HW + q]) * (v[i] - u[i]);
  }
}

__global__ void class_embed_add_kernel(const float* __restrict__ temb, const float* __restrict__ table,
                                       const long* __restrict__ y, float* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, d = i - b * dim;
  out[i] = temb[i] + table[y[b] * dim + d];
}

// one thread per table element: deterministic (no atomics), fixed order over the batch
__global__ void class_embed_bwd_kernel(const float* __restrict__ dout, const long* __restrict__ y,
                                       float* __restrict__ dtable, int B, int dim, int num_classes, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_classes * dim) return;
  const int k = i / dim, d = i - k * dim;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    if (y[b] == k) s += dout[b * dim + d];
  dtable[i] = accumulate ? dtable[i] + s : s;
}

}  // namespace

// v, u: float[B][C][HW]; mask: float[B][HW]; out: float[1]; dv (optional) = grad_scale * d loss / dv;
// work: double[512*2 + 2]
extern "C" int s2s_weighted_mse(const float* v, const float* u, const float* mask, float* dv, float* out,
                                double* work, long B, int C, long HW, float roi_lambda, float grad_scale,
                                void* stream) {
  if (!v || !u || !mask || !out || !work) return S2S_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  long nb = (B * HW + 255) / 256;
  if (nb > LV_BLOCKS) nb = LV_BLOCKS;
  double* sums = work + (long)LV_BLOCKS * 2;
  hipLaunchKernelGGL(lv_reduce_kernel<0>, dim3((int)nb), dim3(256), 0, s, v, u, mask, B, C, HW, roi_lambda, work);
  hipLaunchKernelGGL(lv_finalize_kernel, dim3(1), dim3(64), 0, s, work, (int)nb, 1.0, 1e-8, out, sums);
  if (dv) {
    long gb = (B * C * HW + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(wmse_bwd_kernel, dim3((int)gb), dim3(256), 0, s, v, u, mask, B, C, HW, roi_lambda, sums,
                       grad_scale, dv);
  }
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// pred, truth: float[B][C][HW]; mask: float[B][HW]; out: float[1]; work: double[512*2 + 2]
extern "C" int s2s_charbonnier_roi(const float* pred, const float* truth, const float* mask, float* out, double* work,
                                   long B, int C, long HW, float eps_charb, float eps_area, void* stream) {
  if (!pred || !truth || !mask || !out || !work) return S2S_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return S2S_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  long nb = (B * HW + 255) / 256;
  if (nb > LV_BLOCKS) nb = LV_BLOCKS;
  hipLaunchKernelGGL(lv_reduce_kernel<1>, dim3((int)nb), dim3(256), 0, s, pred, truth, mask, B, C, HW, eps_charb,
                     work);
  hipLaunchKernelGGL(lv_finalize_kernel, dim3(1), dim3(64), 0, s, work, (int)nb, (double)C, (double)eps_area, out,
                     work + (long)LV_BLOCKS * 2);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// out[b][:] = temb[b][:] + table[y[b]][:]; y: int64[B] in [0, num_classes) (checked by the caller)
extern "C" int s2s_class_embed_add(const float* temb, const float* table, const long* y, float* out, int B, int dim,
                                   void* stream) {
  if (!temb || !table || !y || !out) return S2S_ERR_NULL;
  if (B <= 0 || dim <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(class_embed_add_kernel, dim3(cdiv(B * dim, 256)), dim3(256), 0, (hipStream_t)stream, temb, table,
                     y, out, B, dim);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// dtable[k][:] (+)= sum over b with y[b] == k of dout[b][:]
extern "C" int s2s_class_embed_bwd(const float* dout, const long* y, float* dtable, int accumulate, int B, int dim,
                                   int num_classes, void* stream) {
  if (!dout || !y || !dtable) return S2S_ERR_NULL;
  if (B <= 0 || dim <= 0 || num_classes <= 0) return S2S_ERR_SHAPE;
  hipLaunchKernelGGL(class_embed_bwd_kernel, dim3(cdiv(num_classes * dim, 256)), dim3(256), 0, (hipStream_t)stream,
                     dout, y, dtable, B, dim, num_classes, accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
