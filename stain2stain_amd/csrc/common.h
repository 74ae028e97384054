// Shared device helpers for the stain2stain gfx950 kernels.
// Everything here targets CDNA4 (wave64, MFMA, 160 KiB LDS); there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// status codes returned across the C ABI (0 = ok)
#define S2S_OK 0
#define S2S_ERR_SHAPE (-1)
#define S2S_ERR_ALIGN (-2)
#define S2S_ERR_DTYPE (-3)
#define S2S_ERR_LAUNCH (-4)
#define S2S_ERR_NULL (-5)

// dtype tags of the ABI
#define S2S_BF16 0
#define S2S_F32 1

#define S2S_LAUNCH_CHECK()                          \
  do {                                              \
    if (hipGetLastError() != hipSuccess) return S2S_ERR_LAUNCH; \
  } while (0)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 8 consecutive channels of one pixel, as fp32
struct f32x8 {
  float v[8];
};

__device__ __forceinline__ f32x8 load8(const float* p) {
  f32x8 r;
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
  return r;
}
__device__ __forceinline__ f32x8 load8(const bf16_t* p) {
  f32x8 r;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
  return r;
}
__device__ __forceinline__ void store8(float* p, const f32x8& r) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = r.v[i]; b[i] = r.v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const f32x8& r) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16_t)r.v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Opt a kernel into more than 64 KiB of dynamic LDS.  The attribute is a property of the (kernel, device) pair, so a
// process that drives several GPUs has to set it on each of them: `done` is the caller's per-kernel bit mask of the
// devices already served (a benign race: setting the attribute twice is harmless).
static inline int s2s_allow_dyn_lds(const void* kern, int bytes, unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return S2S_ERR_LAUNCH;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && ((__atomic_load_n(done, __ATOMIC_RELAXED) >> dev) & 1ull)) return S2S_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return S2S_ERR_LAUNCH;
  if (tracked) __atomic_fetch_or(done, 1ull << dev, __ATOMIC_RELAXED);
  return S2S_OK;
}
