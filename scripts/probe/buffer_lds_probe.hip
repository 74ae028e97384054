// Probe: buffer_load_dwordx4 ... lds (raw buffer resource, 16 B per lane) on gfx950 -- destination layout (M0 base + lane * 16)
// and zero fill of out-of-range offsets.  Build + run: hipcc --offload-arch=gfx950 -O3 buffer_lds_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void k(const char* x, int nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 2048 / 4; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned off = (63 - threadIdx.x) * 16;                 // lane l fetches piece 63 - l
  if (threadIdx.x % 5 == 0) off = 0x80000000u + threadIdx.x * 16;   // out of range: zero fill expected
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(smem + 1024), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048 / 4; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  std::vector<unsigned> h(256);
  for (int i = 0; i < 256; ++i) h[i] = i;
  char* d; unsigned* o;
  hipMalloc(&d, 1024); hipMalloc(&o, 2048);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 1024, o);
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (r[i] != 0xdeadbeefu) ++bad;          // bytes [0, 1024) untouched
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const unsigned want = (l % 5 == 0) ? 0u : (unsigned)((63 - l) * 4 + j);
      if (r[256 + l * 4 + j] != want) { if (bad < 8) printf("lane %d dword %d: got %u want %u\n", l, j, r[256 + l * 4 + j], want); ++bad; }
    }
  printf(bad ? "PROBE FAIL (%d)\n" : "PROBE OK: lane l -> base + 16 l, out-of-range lanes read zeros\n", bad);
  return bad != 0;
}
