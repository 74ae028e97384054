// Probe: raw buffer store / LDS load with a scalar offset (soffset) on gfx950 -- lanes whose VGPR offset is out of range are
// dropped (store) / zero-filled (load) whatever the scalar offset, in-range lanes land at base + soffset + voffset.
//   hipcc --offload-arch=gfx950 -O3 buffer_store_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(char* y, int nbytes, int soff, const char* x, unsigned* lds_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, nbytes, 0x00020000);
  const int l = threadIdx.x;
  unsigned off = l * 16;
  if (l % 3 == 0) off = 0xffffff00u;
  v4i d = {l, l + 100, l + 200, l + 300};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, off, soff, 0);
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)smem, 16, off, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 256; i += 64) lds_out[i] = ((unsigned*)smem)[i];
}
int main() {
  const int N = 1 << 16, soff = 4096 + 512;
  std::vector<int> h(N / 4, -1), hx(N / 4);
  for (int i = 0; i < N / 4; ++i) hx[i] = i;
  char *d, *x; unsigned* lo;
  hipMalloc(&d, N); hipMalloc(&x, N); hipMalloc(&lo, 1024);
  hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), N, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, N, soff, x, lo);
  std::vector<unsigned> lr(256);
  hipMemcpy(h.data(), d, N, hipMemcpyDeviceToHost);
  hipMemcpy(lr.data(), lo, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < N / 4; ++i) {
    const int b = i * 4 - soff;                      // byte position relative to the scalar offset
    int want = -1;
    if (b >= 0 && b < 64 * 16) { const int l = b / 16, j = (b % 16) / 4; if (l % 3) want = l + 100 * j; }
    if (h[i] != want) { if (bad < 8) printf("store: dword %d got %d want %d\n", i, h[i], want); ++bad; }
  }
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const unsigned want = (l % 3 == 0) ? 0u : (unsigned)((soff + l * 16) / 4 + j);
      if (lr[l * 4 + j] != want) { if (bad < 16) printf("load: lane %d dword %d got %u want %u\n", l, j, lr[l * 4 + j], want); ++bad; }
    }
  printf(bad ? "PROBE FAIL (%d)\n" : "PROBE OK: OOB VGPR offsets dropped / zero-filled with a scalar offset in use\n", bad);
  return bad != 0;
}
