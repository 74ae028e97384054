// Elementwise / loss kernels of the pix2pix G + D step (named by BASELINE.json's
// north_star, absent from the reference repository -- SURVEY.md F1 -- so the semantics are torch's operators, which a
// pix2pix written in the reference's framework would call, and tests/test_pix2pix_gpu.py compares against exactly those):
//
//   p2p_pack_input      NCHW fp32 image pair -> one NHWC tensor padded to 8 channels (the generator's / discriminator's input)
//   p2p_tanh_l1_fwd     fake = tanh(h); the discriminator's input [src | fake | 0 0]; optional NCHW fp32 copy; sum |fake - tgt|
//   p2p_tanh_l1_bwd     dh = (lambda/N sign(fake - tgt) + dD_input[fake channels]) (1 - fake^2)
//   p2p_bce_logits      BCEWithLogits of the PatchGAN logit map against all-ones / all-zeros targets, value + gradient
//   p2p_act_bwd         backward of a LeakyReLU / ReLU that has no norm in front (mask from the stored OUTPUT), fused with
//                       the conv-bias gradient (per-channel sum)
//
// All HBM-bound, 16 B (bf16) / 32 B (fp32) per thread and pixel; T = bf16 throughput mode, T = float parity mode.
#include "common.h"

namespace {

constexpr int NCH = 8;       // the image tensors of this path are padded to 8 channels (one 16-byte piece per pixel in bf16)

// ---- p2p_pack_input -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void p2p_pack_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb,
                                                       T* __restrict__ out, int ldo, int B, long HW) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, p = i - n * HW;
    f32x8 v;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      float t = 0.f;
      if (k < ca) t = a[(n * ca + k) * HW + p];
      else if (k < ca + cb) t = b[(n * cb + (k - ca)) * HW + p];
      v.v[k] = t;
    }
    store8(out + i * ldo, v);
  }
}

// ---- p2p_unpack: the inverse for gradients -- out_nchw[n][c][p] = in[n][p][c0 + c] as fp32 ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void p2p_unpack_kernel(const T* __restrict__ in, int ldi, int c0, int C,
                                                         float* __restrict__ out, int B, long HW) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, p = i - n * HW;
    const f32x8 v = load8(in + i * ldi);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      if (k >= c0 && k < c0 + C) out[(n * C + (k - c0)) * HW + p] = v.v[k];
  }
}

// ---- p2p_tanh_l1_fwd ------------------------------------------------------------------------------------------------
// h: [B][HW][ldh] (C real channels); d_in: [B][HW][ldd] <- [src (C) | tanh(h) (C) | zeros]; fake_nchw (optional);
// lpart[block] = sum |tanh(h) - tgt| in fp64
template <typename T>
__global__ __launch_bounds__(256) void p2p_tanh_l1_fwd_kernel(const T* __restrict__ h, int ldh, const float* __restrict__ src,
                                                              const float* __restrict__ tgt, T* __restrict__ d_in, int ldd,
                                                              float* __restrict__ fake_nchw, double* __restrict__ lpart, int B,
                                                              long HW, int C) {
  const long total = (long)B * HW;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, p = i - n * HW;
    const f32x8 hv = load8(h + i * ldh);
    f32x8 o;
#pragma unroll
    for (int k = 0; k < NCH; ++k) o.v[k] = 0.f;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < C) {
        const float f = tanhf(hv.v[k]);
        const long q = (n * C + k) * HW + p;
        o.v[k] = src[q];
#pragma unroll
        for (int j = 1; j < NCH; ++j)       // o.v[C + k] = f without a run-time register index
          if (j == C + k) o.v[j] = f;
        if (fake_nchw) fake_nchw[q] = f;
        part += fabsf(f - tgt[q]);
      }
    }
    store8(d_in + i * ldd, o);
    acc += (double)part;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) lpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = scale * sum(lpart[0..n))   (one workgroup; fixed order)
__global__ __launch_bounds__(256) void p2p_sum_kernel(const double* __restrict__ lpart, int n, double scale, float* __restrict__ out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += lpart[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * scale);
}

// ---- p2p_tanh_l1_bwd ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void p2p_tanh_l1_bwd_kernel(const T* __restrict__ h, int ldh, const float* __restrict__ tgt,
                                                              const T* __restrict__ gd, int ldg, float l1_scale,
                                                              T* __restrict__ dh, int lddh, int B, long HW, int C) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, p = i - n * HW;
    const f32x8 hv = load8(h + i * ldh);
    f32x8 gv;
    if (gd) gv = load8(gd + i * ldg);
    else {
#pragma unroll
      for (int k = 0; k < NCH; ++k) gv.v[k] = 0.f;
    }
    f32x8 o;
#pragma unroll
    for (int k = 0; k < NCH; ++k) o.v[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < C) {
        const float f = tanhf(hv.v[k]);
        const float d = f - tgt[(n * C + k) * HW + p];
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);          // torch's abs'(0) = 0
        float gk = 0.f;
#pragma unroll
        for (int j = 1; j < NCH; ++j)       // gv.v[C + k] without a run-time register index
          if (j == C + k) gk = gv.v[j];
        o.v[k] = (l1_scale * sgn + gk) * (1.f - f * f);
      }
    }
    store8(dh + i * lddh, o);
  }
}

// ---- p2p_bce_logits -------------------------------------------------------------------------------------------------
// z: [N][HW][ldz], logit = channel 0.  Samples n < n_real have target 1, the others target 0.
//   out[0] = mean over the real samples of softplus(-z), out[1] = mean over the others of softplus(z)
//   dz[n][p][0] = w_real * (sigmoid(z) - 1) | w_fake * sigmoid(z); channels 1..7 = 0
// A PatchGAN logit map is tiny (batch x 30 x 30): one workgroup, fixed summation order.
template <typename T>
__global__ __launch_bounds__(1024) void p2p_bce_kernel(const T* __restrict__ z, int ldz, int n_real, float w_real, float w_fake,
                                                       T* __restrict__ dz, int lddz, float* __restrict__ out, int N, int HW) {
  const long total = (long)N * HW, nr = (long)n_real * HW;
  double a_real = 0.0, a_fake = 0.0;
  constexpr int U = 8;                           // loads of U strides in flight: the loop is latency-, not bandwidth-bound
  for (long base = threadIdx.x; base < total; base += 1024 * U) {
    float xs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * 1024;
      xs[u] = i < total ? to_f32(z[i * ldz]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * 1024;
      if (i >= total) break;
      const float x = xs[u];
      const bool real = i < nr;
      const float sp = fmaxf(real ? -x : x, 0.f) + log1pf(expf(-fabsf(x)));       // softplus(+-x), stable
      const float sg = 1.f / (1.f + expf(-x));
      if (real) a_real += (double)sp; else a_fake += (double)sp;
      if (dz) {
        f32x8 o;
#pragma unroll
        for (int k = 0; k < NCH; ++k) o.v[k] = 0.f;
        o.v[0] = real ? w_real * (sg - 1.f) : w_fake * sg;
        store8(dz + i * lddz, o);
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a_real += __shfl_xor(a_real, m, 64); a_fake += __shfl_xor(a_fake, m, 64); }
  __shared__ double red[2][16];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a_real; red[1][threadIdx.x >> 6] = a_fake; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0, f = 0.0;
    for (int k = 0; k < 16; ++k) { r += red[0][k]; f += red[1][k]; }
    out[0] = nr > 0 ? (float)(r / (double)nr) : 0.f;
    out[1] = total > nr ? (float)(f / (double)(total - nr)) : 0.f;
  }
}

// The same over many workgroups (the softplus / sigmoid arithmetic of a batch-64 PatchGAN map keeps one CU busy for
// 20-40 us): partial sums per workgroup in part[2][gridDim.x], summed in workgroup order by p2p_bce_finish_kernel.
template <typename T>
__global__ __launch_bounds__(256) void p2p_bce_part_kernel(const T* __restrict__ z, int ldz, int n_real, float w_real,
                                                           float w_fake, T* __restrict__ dz, int lddz,
                                                           double* __restrict__ part, int N, int HW) {
  const long total = (long)N * HW, nr = (long)n_real * HW;
  double a_real = 0.0, a_fake = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float x = to_f32(z[i * ldz]);
    const bool real = i < nr;
    const float sp = fmaxf(real ? -x : x, 0.f) + log1pf(expf(-fabsf(x)));
    const float sg = 1.f / (1.f + expf(-x));
    if (real) a_real += (double)sp; else a_fake += (double)sp;
    if (dz) {
      f32x8 o;
#pragma unroll
      for (int k = 0; k < NCH; ++k) o.v[k] = 0.f;
      o.v[0] = real ? w_real * (sg - 1.f) : w_fake * sg;
      store8(dz + i * lddz, o);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a_real += __shfl_xor(a_real, m, 64); a_fake += __shfl_xor(a_fake, m, 64); }
  __shared__ double red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a_real; red[1][threadIdx.x >> 6] = a_fake; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[gridDim.x + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(64) void p2p_bce_finish_kernel(const double* __restrict__ part, int nb, long nr, long total,
                                                            float* __restrict__ out) {
  double r = 0.0, f = 0.0;
  for (int k = threadIdx.x; k < nb; k += 64) { r += part[k]; f += part[nb + k]; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { r += __shfl_xor(r, m, 64); f += __shfl_xor(f, m, 64); }
  if (threadIdx.x == 0) {
    out[0] = nr > 0 ? (float)(r / (double)nr) : 0.f;
    out[1] = total > nr ? (float)(f / (double)(total - nr)) : 0.f;
  }
}

inline int bce_blocks(long total) {
  long nb = (total + 1023) / 1024;
  return (int)(nb < 1 ? 1 : nb > 512 ? 512 : nb);
}

// ---- p2p_act_bwd ----------------------------------------------------------------------------------------------------
// a = lrelu(z) (or relu(z)) stored by the conv epilogue: a > 0 <=> z > 0.  dz = a > 0 ? g + g2 : slope * g  (g2 = the
// gradient wrt the ReLU'd copy in the decoder's concatenation buffer, optional).  part[2][C][blocks]: per-channel sums of
// dz (row 0; row 1 unused) in the layout channel_sum's finalize reads.
__host__ __device__ inline int act_pcb(int C) {
  int p = 1;
  while (p < 32 && p * 8 < C) p <<= 1;
  return p;
}

template <typename T>
__global__ __launch_bounds__(256) void p2p_act_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ g2, int ldg2,
                                                          const T* __restrict__ a, int lda, float slope, T* __restrict__ dz,
                                                          int lddz, float* __restrict__ part, long npix, int C) {
  const int PCB = act_pcb(C), WL = 256 / PCB;
  const int pc = threadIdx.x & (PCB - 1), wl = threadIdx.x / PCB;
  const int c8 = (blockIdx.x * PCB + pc) * 8;
  float s1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s1[k] = 0.f;
  if (c8 < C) {
    for (long p = (long)blockIdx.y * WL + wl; p < npix; p += (long)gridDim.y * WL) {
      const f32x8 av = load8(a + p * lda + c8);
      const f32x8 gv = load8(g + p * ldg + c8);
      f32x8 gw;
      if (g2) gw = load8(g2 + p * ldg2 + c8);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) gw.v[k] = 0.f;
      }
      f32x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o.v[k] = av.v[k] > 0.f ? gv.v[k] + gw.v[k] : slope * gv.v[k];
        s1[k] += o.v[k];
      }
      store8(dz + p * lddz + c8, o);
    }
  }
  if (!part) return;
  __shared__ float red[2304];
  const int rowlen = PCB * 8 + 1;
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wl * rowlen + pc * 8 + k] = s1[k];
  __syncthreads();
  for (int i = threadIdx.x; i < PCB * 8; i += 256) {
    float s = 0.f;
    for (int r = 0; r < WL; ++r) s += red[r * rowlen + i];
    const int c = blockIdx.x * PCB * 8 + i;
    if (c < C) part[(long)c * gridDim.y + blockIdx.y] = s;
  }
}

// dbias[c] (+)= sum of the row of partials (one wave per channel, fp64)
__global__ __launch_bounds__(256) void p2p_rowsum_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out,
                                                         int accumulate) {
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= C) return;
  double acc = 0.0;
  for (int k = threadIdx.x & 63; k < nblk; k += 64) acc += (double)part[i * nblk + k];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) out[i] = accumulate ? out[i] + (float)acc : (float)acc;
}

inline int act_blocks(long npix, int C) {
  const int pcb = act_pcb(C), wl = 256 / pcb, groups = cdiv(C / 8, pcb);
  long nb = cdiv(4096, groups);
  const long most = (npix + wl - 1) / wl;
  if (nb > most) nb = most;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}

inline unsigned ew_blocks(long total) {
  long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

}  // namespace

extern "C" int s2s_p2p_pack_input(int dtype, const float* a_nchw, int ca, const float* b_nchw, int cb, void* out, int ldo,
                                  int B, int H, int W, void* stream) {
  if (!a_nchw || !out) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || ca <= 0 || cb < 0 || ca + cb > NCH || (cb > 0 && !b_nchw) || (ldo % 8) || ldo < NCH) return S2S_ERR_SHAPE;
  if ((uintptr_t)out & 15) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long HW = (long)H * W;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(p2p_pack_kernel<bf16_t>, dim3(ew_blocks(B * HW)), dim3(256), 0, s, a_nchw, ca, b_nchw, cb, (bf16_t*)out, ldo, B, HW);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(p2p_pack_kernel<float>, dim3(ew_blocks(B * HW)), dim3(256), 0, s, a_nchw, ca, b_nchw, cb, (float*)out, ldo, B, HW);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_p2p_unpack(int dtype, const void* in, int ldi, int c0, int C, float* out_nchw, int B, int H, int W,
                              void* stream) {
  if (!in || !out_nchw) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || c0 < 0 || c0 + C > NCH || (ldi % 8) || ldi < NCH) return S2S_ERR_SHAPE;
  if ((uintptr_t)in & 15) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long HW = (long)H * W;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(p2p_unpack_kernel<bf16_t>, dim3(ew_blocks(B * HW)), dim3(256), 0, s, (const bf16_t*)in, ldi, c0, C, out_nchw, B, HW);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(p2p_unpack_kernel<float>, dim3(ew_blocks(B * HW)), dim3(256), 0, s, (const float*)in, ldi, c0, C, out_nchw, B, HW);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_p2p_tanh_l1_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return S2S_ERR_SHAPE;
  long nb = ((long)B * H * W + 255) / 256;
  if (nb > 2048) nb = 2048;
  return (int)nb;
}

// l1_out[0] = mean |tanh(h) - tgt| over B*C*H*W; work: double[s2s_p2p_tanh_l1_blocks()]
extern "C" int s2s_p2p_tanh_l1_fwd(int dtype, const void* h, int ldh, const float* src_nchw, const float* tgt_nchw, void* d_in,
                                   int ldd, float* fake_nchw, float* l1_out, void* work, int B, int H, int W, int C,
                                   void* stream) {
  if (!h || !src_nchw || !tgt_nchw || !d_in || !l1_out || !work) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4 || (ldh % 8) || (ldd % 8) || ldh < NCH || ldd < NCH) return S2S_ERR_SHAPE;
  if (((uintptr_t)h & 15) || ((uintptr_t)d_in & 15)) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long HW = (long)H * W;
  const int nb = s2s_p2p_tanh_l1_blocks(B, H, W);
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(p2p_tanh_l1_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)h, ldh, src_nchw, tgt_nchw,
                       (bf16_t*)d_in, ldd, fake_nchw, (double*)work, B, HW, C);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(p2p_tanh_l1_fwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)h, ldh, src_nchw, tgt_nchw,
                       (float*)d_in, ldd, fake_nchw, (double*)work, B, HW, C);
  else return S2S_ERR_DTYPE;
  hipLaunchKernelGGL(p2p_sum_kernel, dim3(1), dim3(256), 0, s, (const double*)work, nb, 1.0 / ((double)B * C * HW), l1_out);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// dh[c] = (l1_scale * sign(tanh(h) - tgt) + gd[C + c]) * (1 - tanh(h)^2), c < C; the padding channels get 0.
// gd (optional): gradient wrt the discriminator's input [src | fake | 0 0], NHWC in the activation dtype.
extern "C" int s2s_p2p_tanh_l1_bwd(int dtype, const void* h, int ldh, const float* tgt_nchw, const void* gd, int ldg,
                                   float l1_scale, void* dh, int lddh, int B, int H, int W, int C, void* stream) {
  if (!h || !tgt_nchw || !dh) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4 || (ldh % 8) || (ldg % 8) || (lddh % 8) || ldh < NCH || lddh < NCH || (gd && ldg < NCH)) return S2S_ERR_SHAPE;
  if (((uintptr_t)h & 15) || ((uintptr_t)gd & 15) || ((uintptr_t)dh & 15)) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long HW = (long)H * W;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(p2p_tanh_l1_bwd_kernel<bf16_t>, dim3(ew_blocks(B * HW)), dim3(256), 0, s, (const bf16_t*)h, ldh, tgt_nchw,
                       (const bf16_t*)gd, ldg, l1_scale, (bf16_t*)dh, lddh, B, HW, C);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(p2p_tanh_l1_bwd_kernel<float>, dim3(ew_blocks(B * HW)), dim3(256), 0, s, (const float*)h, ldh, tgt_nchw,
                       (const float*)gd, ldg, l1_scale, (float*)dh, lddh, B, HW, C);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_p2p_bce_logits(int dtype, const void* z, int ldz, int n_real, float w_real, float w_fake, void* dz,
                                  int lddz, float* out2, int N, int HW, void* stream) {
  if (!z || !out2) return S2S_ERR_NULL;
  if (N <= 0 || HW <= 0 || n_real < 0 || n_real > N || ldz <= 0 || (dz && ((lddz % 8) || lddz < NCH))) return S2S_ERR_SHAPE;
  if ((uintptr_t)dz & 15) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(p2p_bce_kernel<bf16_t>, dim3(1), dim3(1024), 0, s, (const bf16_t*)z, ldz, n_real, w_real, w_fake,
                       (bf16_t*)dz, lddz, out2, N, HW);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(p2p_bce_kernel<float>, dim3(1), dim3(1024), 0, s, (const float*)z, ldz, n_real, w_real, w_fake,
                       (float*)dz, lddz, out2, N, HW);
  else return S2S_ERR_DTYPE;
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_p2p_bce_blocks(int N, int HW) {
  if (N <= 0 || HW <= 0) return S2S_ERR_SHAPE;
  return bce_blocks((long)N * HW);
}

// s2s_p2p_bce_logits over s2s_p2p_bce_blocks(N, HW) workgroups; work: double[2 * s2s_p2p_bce_blocks(N, HW)].
extern "C" int s2s_p2p_bce_logits_w(int dtype, const void* z, int ldz, int n_real, float w_real, float w_fake, void* dz,
                                    int lddz, float* out2, double* work, int N, int HW, void* stream) {
  if (!z || !out2 || !work) return S2S_ERR_NULL;
  if (N <= 0 || HW <= 0 || n_real < 0 || n_real > N || ldz <= 0 || (dz && ((lddz % 8) || lddz < NCH))) return S2S_ERR_SHAPE;
  if (((uintptr_t)dz & 15) || ((uintptr_t)work & 7)) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * HW;
  const int nb = bce_blocks(total);
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(p2p_bce_part_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)z, ldz, n_real, w_real, w_fake,
                       (bf16_t*)dz, lddz, work, N, HW);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(p2p_bce_part_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)z, ldz, n_real, w_real, w_fake,
                       (float*)dz, lddz, work, N, HW);
  else return S2S_ERR_DTYPE;
  hipLaunchKernelGGL(p2p_bce_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)work, nb, (long)n_real * HW, total, out2);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_p2p_act_bwd_blocks(long npix, int C) {
  if (npix <= 0 || C <= 0 || (C % 8)) return S2S_ERR_SHAPE;
  return act_blocks(npix, C);
}

// dz = a > 0 ? g + g2 : slope * g over [npix][C] (g2 optional); dbias (optional) (+)= per-channel sum of dz.
// work: float[C * s2s_p2p_act_bwd_blocks()].
extern "C" int s2s_p2p_act_bwd(int dtype, const void* g, int ldg, const void* g2, int ldg2, const void* a, int lda,
                               float slope, void* dz, int lddz, float* work, float* dbias, int accumulate, long npix, int C,
                               void* stream) {
  if (!g || !a || !dz || (dbias && !work)) return S2S_ERR_NULL;
  if (npix <= 0 || C <= 0 || (C % 8) || (ldg % 8) || (lda % 8) || (lddz % 8) || (g2 && (ldg2 % 8))) return S2S_ERR_SHAPE;
  if (((uintptr_t)g & 15) || ((uintptr_t)g2 & 15) || ((uintptr_t)a & 15) || ((uintptr_t)dz & 15)) return S2S_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int nb = act_blocks(npix, C);
  const dim3 grid(cdiv(C / 8, act_pcb(C)), nb);
  float* part = dbias ? work : nullptr;
  if (dtype == S2S_BF16)
    hipLaunchKernelGGL(p2p_act_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)g, ldg, (const bf16_t*)g2, ldg2,
                       (const bf16_t*)a, lda, slope, (bf16_t*)dz, lddz, part, npix, C);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(p2p_act_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)g, ldg, (const float*)g2, ldg2,
                       (const float*)a, lda, slope, (float*)dz, lddz, part, npix, C);
  else return S2S_ERR_DTYPE;
  if (dbias)
    hipLaunchKernelGGL(p2p_rowsum_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, part, nb, C, dbias, accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
