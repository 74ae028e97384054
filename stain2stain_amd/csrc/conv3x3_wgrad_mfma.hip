// Weight gradient of the 3x3 / pad 1 convolution, NHWC activations, MFMA (gfx950).
//
// Replaces autograd's conv2d weight-gradient for the reference's DoubleConv layers
// (src/models/components/shared_encoder.py:15,18; task_decoders.py:15,18).
//
//   dW[tap][co][ci] = sum_{pixel} dY[pixel][co] * X[pixel (+) tap][ci]
//
// As a GEMM the reduction index is the PIXEL, which is the slow axis of both NHWC operands, so the
// MFMA fragments (8 consecutive k per lane) are columns of the staged LDS images.  They are read
// with ds_read_b64_tr_b16 (CDNA4 transposed LDS read): the images stay [pixel][32 channels]
// (64-B rows, which also makes the transposed reads bank-conflict free: 4 rows x 2 column groups
// of one half-wave cover the 256-B bank row exactly once), and the tap is a constant row offset
// into the (TH+2)x(TW+2) halo image, so one staged halo serves all nine taps.
//
//   * workgroup = 4 waves = 64 co x 64 ci x 9 taps; each wave owns a 32x32 (co,ci) block for all
//     nine taps (9 x 16 accumulator registers), so a dY fragment is read once per 9 MFMAs.
//   * split-K over pixel tiles: blockIdx.z walks tiles z, z+S, ...; partial sums go to
//     part[S][9][Cout][Cin] (fp32) and s2s_conv3x3_wgrad_reduce folds them, deterministically, into
//     the OIHW fp32 gradient the optimiser sees.
//   * T = float: three-way bf16 split of both operands, 6 MFMAs per product (see conv3x3_mfma.hip).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct WgradArgs {
  const void* dy;
  const void* x0;
  const void* x1;
  float* part;
  int lddy, Cout, ld0, c0, ld1, c1;
  int B, H, W, tilesY, tilesX, ntiles, S;
  int lgc;   // conv2x2_wgrad_dma_kernel<.., S2D = true>: log2 of the real channel count of the plain X tensor
  int lgh, lgw;   // conv2x2_wgrad_flat_kernel: log2 of the (output) map size
  int xcd;        // conv3x3_wgrad_dma_kernel: 1 = XCD-aware workgroup order (an XCD takes a contiguous range of the
                  // (pixel split, co tile, ci tile) sequence, so the workgroups that stream the same dY / X tiles share an L2)
};

namespace {

template <typename T> struct WPiece;
template <> struct WPiece<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.0f;
  }
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void to_lds(char* hi, int, int off) const { *reinterpret_cast<bf16x8*>(hi + off) = v; }
};
template <> struct WPiece<float> {
  f32x4 a, b;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = 0.f; b[i] = 0.f; }
  }
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void to_lds(char* hi, int img_stride, int off) const {
    bf16x8 h, m, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = i < 4 ? a[i] : b[i - 4];
      h[i] = (bf16_t)x;
      const float r1 = x - (float)h[i];
      m[i] = (bf16_t)r1;
      l[i] = (bf16_t)(r1 - (float)m[i]);
    }
    *reinterpret_cast<bf16x8*>(hi + off) = h;
    *reinterpret_cast<bf16x8*>(hi + img_stride + off) = m;
    *reinterpret_cast<bf16x8*>(hi + 2 * img_stride + off) = l;
  }
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// 8 consecutive-k elements of one column: two transposed 4x16 block reads (rows +0..3, +4..7)
__device__ __forceinline__ bf16x8 tr_frag(const char* p_rows0, int row_stride4) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p_rows0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p_rows0 + row_stride4));
  bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = l4[i]; r[4 + i] = h4[i]; }
  return r;
}

template <typename T, int TH, int TW>
__global__ __launch_bounds__(256, (std::is_same<T, float>::value ? 1 : 2)) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr bool SPLIT = std::is_same<T, float>::value;
  constexpr int NIMG = SPLIT ? 3 : 1;
  constexpr int NPX = TH * TW;                 // 128
  constexpr int HW_ = TW + 2, HALO = (TH + 2) * HW_;
  constexpr int DY_BYTES = 2 * NPX * 64;       // two 32-channel images
  constexpr int X_BYTES = 2 * HALO * 64;
  constexpr int DY_PIECES = NPX * 8, DY_IT = (DY_PIECES + 255) / 256;
  constexpr int X_PIECES = HALO * 8, X_IT = (X_PIECES + 255) / 256;
  constexpr int KSTEPS = NPX / 16;
  static_assert(TW % 16 == 0, "a k-step is 16 consecutive pixels of one tile row");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsDY = smem;                          // [NIMG][2][NPX][64B]
  char* const ldsX = smem + NIMG * DY_BYTES;         // [NIMG][2][HALO][64B]

  const T* __restrict__ dy = static_cast<const T*>(a.dy);
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave >> 1, wci = wave & 1;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int cin = a.c0 + a.c1;

  // transposed-read lane geometry (see header comment)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + p * 4) * 2;  // bytes, k-row 0 of the step

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  WPiece<T> dreg[DY_IT];
  WPiece<T> xreg[X_IT];

  auto load_tile = [&](int tile) {
    int bt = tile;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int img = bt / a.tilesY;
    const int y0 = ty * TH, xs = tx * TW;
#pragma unroll
    for (int i = 0; i < DY_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      const int py = px / TW, pxx = px - py * TW;
      const int gy = y0 + py, gx = xs + pxx, co = co0 + pc * 8;
      dreg[i].zero();
      if (idx < DY_PIECES && gy < a.H && gx < a.W && co < a.Cout)
        dreg[i].load(dy + ((long)(img * a.H + gy) * a.W + gx) * a.lddy + co);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      const int hy = px / HW_, hx = px - hy * HW_;
      const int gy = y0 - 1 + hy, gx = xs - 1 + hx, ci = ci0 + pc * 8;
      xreg[i].zero();
      if (idx < X_PIECES && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const long pix = (long)(img * a.H + gy) * a.W + gx;
        if (ci < a.c0) xreg[i].load(x0 + pix * a.ld0 + ci);
        else if (ci < cin) xreg[i].load(x1 + pix * a.ld1 + (ci - a.c0));
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < DY_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      if (idx < DY_PIECES)
        dreg[i].to_lds(ldsDY, DY_BYTES, (pc >> 2) * (NPX * 64) + px * 64 + (pc & 3) * 16);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      if (idx < X_PIECES)
        xreg[i].to_lds(ldsX, X_BYTES, (pc >> 2) * (HALO * 64) + px * 64 + (pc & 3) * 16);
    }
  };

  const char* const Ahi = ldsDY + wco * (NPX * 64) + frag_off;
  const char* const Bhi = ldsX + wci * (HALO * 64) + frag_off;

  int tile = blockIdx.z;
  if (tile < a.ntiles) load_tile(tile);
  for (; tile < a.ntiles; tile += a.S) {
    __syncthreads();  // everyone is done reading the previous tile
    store_tile();
    __syncthreads();
    if (tile + a.S < a.ntiles) load_tile(tile + a.S);  // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int m0 = ks * 16;
      const int py = m0 / TW, px = m0 - py * TW;
      const bf16x8 af = tr_frag(Ahi + m0 * 64, 4 * 64);
      bf16x8 am, al;
      if constexpr (SPLIT) {
        am = tr_frag(Ahi + DY_BYTES + m0 * 64, 4 * 64);
        al = tr_frag(Ahi + 2 * DY_BYTES + m0 * 64, 4 * 64);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int hoff = ((py + tap / 3) * HW_ + px + tap % 3) * 64;
        const bf16x8 bfr = tr_frag(Bhi + hoff, 4 * 64);
        if constexpr (SPLIT) {
          const bf16x8 bm = tr_frag(Bhi + X_BYTES + hoff, 4 * 64);
          const bf16x8 bl = tr_frag(Bhi + 2 * X_BYTES + hoff, 4 * 64);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bfr, acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bl, acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bfr, acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bm, acc[tap], 0, 0, 0);
        }
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tap], 0, 0, 0);
      }
    }
  }

  // ---- partial slab: part[z][tap][co][ci] ----
  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = co0 + wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      if (co < a.Cout && ci < cin)
        a.part[(((long)blockIdx.z * 9 + tap) * a.Cout + co) * cin + ci] = acc[tap][j];
    }
  }
}

// =========================================================================================================
// bf16 v2: the same computation with LDS-DMA staging (global_load_lds_dwordx4) and double-buffered LDS.
// The [pixel][32 ch] images are already the linear layout a DMA wave-instruction writes (16 rows x 64 B),
// so nothing changes on the read side; out-of-range rows fetch a zero page so that every wave issues the
// same 10 DMAs per tile.  One barrier per pixel tile; the next tile's DMAs fly during the current MFMAs.
// =========================================================================================================
__device__ __attribute__((aligned(64))) char g_wgrad_zero_page[64];
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// LDS-DMA issued from inline asm: hipcc's waitcnt pass does not see it, so it cannot put a conservative
// s_waitcnt vmcnt(0) in front of the transposed LDS reads that follow (it does for the builtin form, which
// serialises load and compute); completion is tracked by the explicit vmcnt waits below.  M0 carries the
// wave-uniform LDS destination and is restored inside the same statement.
__device__ __forceinline__ void dma16_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// (Round 4 also tried the tile's 72 MFMAs as an explicit software pipeline -- X fragment of MFMA i + 5 read behind MFMA i,
// sched_group_barrier-pinned, counted lgkmcnt(8..10) in front of every MFMA: 99-110 us against 82 us per 77-GFLOP launch and
// 7.15-7.19 against 7.11-7.12 ms per step; the compiler's own grouping of four reads per three MFMAs stays.)
// (Round 4 also tried a start-up phase offset per workgroup, as conv3x3_stage_kernel has: nothing gained here, 7.13 -> 7.16
// ... 7.24 ms per step for 0.4 ... 1.8 us of offset per phase -- two workgroups per CU already drift apart.)
// A workgroup accumulates all nine taps (one dY fragment feeds nine MFMAs).  (Rounds 2-3 also built the kernel rows over
// three workgroups, over the three 4-wave teams of a 12-wave workgroup, a two-team pixel split and a 16x16x32-MFMA form:
// all parity-green, all 8-18 % slower -- DESIGN.md section 3.2 -- and deleted in round 4.)
template <int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_dma_kernel(WgradArgs a) {
  using T = bf16_t;
  constexpr int NW = 4, NT = 9;
  constexpr int NPX = TH * TW;                             // 128
  constexpr int HW_ = TW + 2, HR = TH + 2, HALO = HR * HW_;
  constexpr int XROWS = (HALO + 31) / 32 * 32;             // 192 (whole DMA groups, 4 waves x 2 halves)
  constexpr int DY_BYTES = 2 * NPX * 64, X_BYTES = 2 * XROWS * 64;
  constexpr int DYGRP = 2 * NPX / 16, XGRP = 2 * XROWS / 16;   // 16-row DMA groups per tile (16 / 24)
  constexpr int DYG = DYGRP / NW;                          // dY DMA instr per wave (4)
  constexpr int XG = XGRP / NW;                            // halo DMA instr per wave (6)
  constexpr int BUF = DY_BYTES + X_BYTES;
  constexpr int KSTEPS = NPX / 16;
  static_assert(DYGRP % 4 == 0 && XGRP % 4 == 0, "DMA groups split evenly over 4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF]
  const T* __restrict__ dy = static_cast<const T*>(a.dy);
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = (wave & 3) >> 1, wci = wave & 1;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {     // workgroups go to the 8 XCDs round-robin by linear id: XCD x takes logical ids [x N/8, (x+1) N/8)
    const unsigned L = bx + gridDim.x * (by + gridDim.y * bz), per = gridDim.x * gridDim.y * gridDim.z / 8;
    unsigned Lq = (L & 7) * per + (L >> 3);
    bx = Lq % gridDim.x; Lq /= gridDim.x;
    by = Lq % gridDim.y; bz = Lq / gridDim.y;
  }
  const int ci0 = bx * 64, co0 = by * 64;
  const int cin = a.c0 + a.c1;
  const int drow = lane >> 2, dslot = lane & 3;
  const int zsplit = (int)bz;

  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + p * 4) * 2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  // Per-lane, tile-invariant parts of the DMA source addresses (which pixel of the tile / halo, which tensor and
  // channel); per tile only a wave-uniform pixel base changes.  Interior tiles (the whole halo inside the image)
  // take a path with no per-lane bounds tests: the address arithmetic otherwise costs a third of the loop's issue
  // slots (measured: 3.3 us per 128-pixel tile against 2.2 us of MFMA for the two waves of a SIMD).
  int dy_off[DYG], dy_py[DYG], dy_px[DYG];
  bool dy_ok[DYG];
#pragma unroll
  for (int j = 0; j < DYG; ++j) {
    const int grp = wave + NW * j;                 // half = grp / 8, rows (grp % 8) * 16 ..
    const int half = grp / (NPX / 16), px = (grp % (NPX / 16)) * 16 + drow;
    const int co = co0 + half * 32 + dslot * 8;
    dy_py[j] = px / TW; dy_px[j] = px - dy_py[j] * TW;
    dy_ok[j] = co < a.Cout;
    dy_off[j] = (dy_py[j] * a.W + dy_px[j]) * a.lddy + co;
  }
  const T* x_ptr[XG];
  int x_off[XG], x_ld[XG], x_hy[XG], x_hx[XG];
#pragma unroll
  for (int j = 0; j < XG; ++j) {
    const int grp = wave + NW * j;
    const int half = grp / (XROWS / 16), px = (grp % (XROWS / 16)) * 16 + drow;
    const int ci = ci0 + half * 32 + dslot * 8;
    x_hy[j] = px / HW_; x_hx[j] = px - x_hy[j] * HW_;
    x_ptr[j] = nullptr; x_ld[j] = 0;
    if (px < HALO) {
      if (ci < a.c0) { x_ptr[j] = x0 + ci; x_ld[j] = a.ld0; }
      else if (ci < cin) { x_ptr[j] = x1 + (ci - a.c0); x_ld[j] = a.ld1; }
    }
    x_off[j] = ((x_hy[j] - 1) * a.W + (x_hx[j] - 1)) * x_ld[j];
  }

  auto dma_tile = [&](int tile, int buf) {
    int bt = tile;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int img = bt / a.tilesY;
    const int y0 = ty * TH, xs = tx * TW;
    const long pixbase = (long)(img * a.H + y0) * a.W + xs;           // wave-uniform
    const bool interior = y0 >= 1 && y0 + TH + 1 <= a.H && xs >= 1 && xs + TW + 1 <= a.W;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + buf * BUF;
    if (interior) {
#pragma unroll
      for (int j = 0; j < DYG; ++j) {
        const void* src = dy_ok[j] ? (const void*)(dy + pixbase * a.lddy + dy_off[j]) : (const void*)g_wgrad_zero_page;
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + (wave + NW * j) * 1024));
      }
#pragma unroll
      for (int j = 0; j < XG; ++j) {
        const void* src = x_ptr[j] ? (const void*)(x_ptr[j] + pixbase * x_ld[j] + x_off[j]) : (const void*)g_wgrad_zero_page;
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + DY_BYTES + (wave + NW * j) * 1024));
      }
    } else {
#pragma unroll
      for (int j = 0; j < DYG; ++j) {
        const int gy = y0 + dy_py[j], gx = xs + dy_px[j];
        const void* src = g_wgrad_zero_page;
        if (gy < a.H && gx < a.W && dy_ok[j]) src = dy + pixbase * a.lddy + dy_off[j];
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + (wave + NW * j) * 1024));
      }
#pragma unroll
      for (int j = 0; j < XG; ++j) {
        const int gy = y0 - 1 + x_hy[j], gx = xs - 1 + x_hx[j];
        const void* src = g_wgrad_zero_page;
        if (x_ptr[j] && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) src = x_ptr[j] + pixbase * x_ld[j] + x_off[j];
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + DY_BYTES + (wave + NW * j) * 1024));
      }
    }
  };

  int tile = zsplit;
  int buf = 0;
  if (tile < a.ntiles) dma_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; tile < a.ntiles; tile += a.S) {
    if (tile + a.S < a.ntiles) dma_tile(tile + a.S, buf ^ 1);   // lands during the MFMAs below
    const char* const Ahi = smem + buf * BUF + wco * (NPX * 64) + frag_off;
    const char* const Bhi = smem + buf * BUF + DY_BYTES + wci * (XROWS * 64) + frag_off;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int m0 = ks * 16;
      const int py = m0 / TW, px = m0 - py * TW;
      const bf16x8 af = tr_frag(Ahi + m0 * 64, 4 * 64);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int hoff = ((py + t / 3) * HW_ + px + t % 3) * 64;
        const bf16x8 bfr = tr_frag(Bhi + hoff, 4 * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
      }
    }
    // every MFMA (hence every LDS read of this buffer) is issued before the barrier: the next iteration's DMA
    // overwrites this buffer's sibling, the one after it this buffer
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }

  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = co0 + wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      if (co < a.Cout && ci < cin)
        a.part[(((long)zsplit * 9 + t) * a.Cout + co) * cin + ci] = acc[t][j];
    }
  }
}

// grad[co][ci][tap] (+)= sum_z part[z][tap][co][ci].  A workgroup owns 64 consecutive (co,ci) pairs; its ZG
// waves take every ZG-th split (reads stay coalesced along ci), the sums meet in LDS and leave as 64*9
// contiguous floats of the OIHW gradient.  ZG = 16 for the narrow layers: a 64 x 64 weight has only 64 such
// workgroups but up to 400 splits to read, so the parallelism has to come from inside the workgroup.
template <int ZG>
__global__ __launch_bounds__(64 * ZG) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ grad, int S, int Cout,
                                    int Cin, int accumulate) {
  __shared__ float tile[ZG][9][65];
  const long n = (long)Cout * Cin;
  const int il = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + il;  // (co, ci)
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  if (i < n) {
    for (int z = zg; z < S; z += ZG)
#pragma unroll
      for (int t = 0; t < 9; ++t) s[t] += part[((long)z * 9 + t) * n + i];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) tile[zg][t][il] = s[t];
  __syncthreads();
  const long base = (long)blockIdx.x * 64 * 9;
  for (int k = threadIdx.x; k < 576; k += 64 * ZG) {
    const long o = base + k;
    if (o < n * 9) {
      const int t = k % 9, j = k / 9;
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < ZG; g += 4) v += (tile[g][t][j] + tile[g + 1][t][j]) + (tile[g + 2][t][j] + tile[g + 3][t][j]);
      grad[o] = accumulate ? grad[o] + v : v;
    }
  }
}

template <typename T, int TH, int TW>
int launch_wgrad(WgradArgs& a, hipStream_t s) {
  constexpr int NIMG = std::is_same<T, float>::value ? 3 : 1;
  constexpr int lds = NIMG * (2 * TH * TW * 64 + 2 * (TH + 2) * (TW + 2) * 64);
  a.tilesY = cdiv(a.H, TH);
  a.tilesX = cdiv(a.W, TW);
  a.ntiles = a.B * a.tilesY * a.tilesX;
  if (a.S > a.ntiles) return S2S_ERR_SHAPE;
  auto kern = conv3x3_wgrad_kernel<T, TH, TW>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(cdiv(a.c0 + a.c1, 64), cdiv(a.Cout, 64), a.S);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

template <int TH, int TW>
int launch_wgrad_dma(WgradArgs& a, hipStream_t s) {
  constexpr int XROWS = ((TH + 2) * (TW + 2) + 31) / 32 * 32;
  constexpr int lds = 2 * (2 * TH * TW * 64 + 2 * XROWS * 64);
  a.tilesY = cdiv(a.H, TH);
  a.tilesX = cdiv(a.W, TW);
  a.ntiles = a.B * a.tilesY * a.tilesX;
  if (a.S > a.ntiles) return S2S_ERR_SHAPE;
  auto kern = conv3x3_wgrad_dma_kernel<TH, TW>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(cdiv(a.c0 + a.c1, 64), cdiv(a.Cout, 64), a.S);
  static const int xcd_aware = [] { const char* e = getenv("S2S_WGRAD_XCD"); return e ? atoi(e) : 1; }();
  a.xcd = xcd_aware && ((long)grid.x * grid.y * grid.z) % 8 == 0;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// =========================================================================================================
// 2x2-tap variant: weight gradient of conv2x2_dma16_kernel's pad = 0 form,
//   dW[a*2+b][co][k] = sum_{n,i,j} dY[n][i][j][co] * X[n][i+a][j+b][k],   X is (H+1) x (W+1), dY is H x W,
// i.e. of the 4x4 stride-2 convolution on the space-to-depth image.  Same staging (LDS-DMA, double-buffered, one
// barrier per pixel tile), same transposed fragment reads, four accumulators instead of nine; the halo is
// (TH+1) x (TW+1) and is never out of range for a tile inside the output.
// =========================================================================================================
// S2D: X is given as the PLAIN tensor [B][2H][2W][C] (a.c0 = 4 C virtual channels, C = 1 << a.lgc) and the loader does the
// space-to-depth in its DMA addresses, as convkxk_dma16_kernel's MODE 1 does: channel (r*2+s)*C + c of cell (p, q) is
// channel c of pixel (2p + r - 1, 2q + s - 1), zero outside.  A lane's channel piece is fixed for the whole kernel, so
// (r, s, c) are per-lane constants.
template <int TH, int TW, bool S2D = false>
__global__ __launch_bounds__(256, 2) void conv2x2_wgrad_dma_kernel(WgradArgs a) {
  using T = bf16_t;
  constexpr int NT = 4;
  constexpr int NPX = TH * TW;
  constexpr int HW_ = TW + 1, HR = TH + 1, HALO = HR * HW_;
  constexpr int XROWS = (HALO + 31) / 32 * 32;
  constexpr int DY_BYTES = 2 * NPX * 64, X_BYTES = 2 * XROWS * 64;
  constexpr int DYG = 2 * NPX / 16 / 4;
  constexpr int XG = 2 * XROWS / 16 / 4;
  constexpr int BUF = DY_BYTES + X_BYTES;
  constexpr int KSTEPS = NPX / 16;
  static_assert((2 * NPX / 16) % 4 == 0 && (2 * XROWS / 16) % 4 == 0, "DMA groups split evenly over 4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF]
  const T* __restrict__ dy = static_cast<const T*>(a.dy);
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wci = wave & 1;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int cin = a.c0;
  const int Hi = a.H + 1, Wi = a.W + 1;
  const int drow = lane >> 2, dslot = lane & 3;
  const int zsplit = (int)blockIdx.z;

  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + p * 4) * 2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  int dy_off[DYG], dy_py[DYG], dy_px[DYG];
  bool dy_ok[DYG];
#pragma unroll
  for (int j = 0; j < DYG; ++j) {
    const int grp = wave + 4 * j;
    const int half = grp / (NPX / 16), px = (grp % (NPX / 16)) * 16 + drow;
    const int co = co0 + half * 32 + dslot * 8;
    dy_py[j] = px / TW; dy_px[j] = px - dy_py[j] * TW;
    dy_ok[j] = co < a.Cout;
    dy_off[j] = (dy_py[j] * a.W + dy_px[j]) * a.lddy + co;
  }
  const T* x_ptr[XG];
  int x_off[XG], x_hy[XG], x_hx[XG];
#pragma unroll
  for (int j = 0; j < XG; ++j) {
    const int grp = wave + 4 * j;
    const int half = grp / (XROWS / 16), px = (grp % (XROWS / 16)) * 16 + drow;
    const int ci = ci0 + half * 32 + dslot * 8;
    x_hy[j] = px / HW_; x_hx[j] = px - x_hy[j] * HW_;
    if (S2D) {
      const int rs = ci >> a.lgc;
      x_ptr[j] = (px < HALO && ci < cin) ? x0 + (ci & ((1 << a.lgc) - 1)) : nullptr;
      x_hy[j] = 2 * x_hy[j] + (rs >> 1) - 1;            // plain-pixel offsets relative to (2 y0, 2 xs)
      x_hx[j] = 2 * x_hx[j] + (rs & 1) - 1;
      x_off[j] = 0;
    } else {
      x_ptr[j] = (px < HALO && ci < cin) ? x0 + ci : nullptr;
      x_off[j] = (x_hy[j] * Wi + x_hx[j]) * a.ld0;
    }
  }

  auto dma_tile = [&](int tile, int buf) {
    int bt = tile;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int img = bt / a.tilesY;
    const int y0 = ty * TH, xs = tx * TW;
    const long pixbase = (long)(img * a.H + y0) * a.W + xs;           // dY (output) pixel, wave-uniform
    const long xbase = (long)(img * Hi + y0) * Wi + xs;               // X (input) pixel of the halo origin
    const bool interior = !S2D && y0 + TH <= a.H && xs + TW <= a.W;   // then the halo is inside the input too
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + buf * BUF;
    if (interior) {
#pragma unroll
      for (int j = 0; j < DYG; ++j) {
        const void* src = dy_ok[j] ? (const void*)(dy + pixbase * a.lddy + dy_off[j]) : (const void*)g_wgrad_zero_page;
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + (wave + 4 * j) * 1024));
      }
#pragma unroll
      for (int j = 0; j < XG; ++j) {
        const void* src = x_ptr[j] ? (const void*)(x_ptr[j] + xbase * a.ld0 + x_off[j]) : (const void*)g_wgrad_zero_page;
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + DY_BYTES + (wave + 4 * j) * 1024));
      }
    } else {
#pragma unroll
      for (int j = 0; j < DYG; ++j) {
        const int gy = y0 + dy_py[j], gx = xs + dy_px[j];
        const void* src = g_wgrad_zero_page;
        if (gy < a.H && gx < a.W && dy_ok[j]) src = dy + pixbase * a.lddy + dy_off[j];
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + (wave + 4 * j) * 1024));
      }
#pragma unroll
      for (int j = 0; j < XG; ++j) {
        const void* src = g_wgrad_zero_page;
        if (S2D) {
          const int py = 2 * y0 + x_hy[j], px = 2 * xs + x_hx[j];
          if (x_ptr[j] && (unsigned)py < (unsigned)(2 * a.H) && (unsigned)px < (unsigned)(2 * a.W))
            src = x_ptr[j] + ((long)(img * 2 * a.H + py) * (2 * a.W) + px) * a.ld0;
        } else {
          const int gy = y0 + x_hy[j], gx = xs + x_hx[j];
          if (x_ptr[j] && gy < Hi && gx < Wi) src = x_ptr[j] + xbase * a.ld0 + x_off[j];
        }
        dma16_asm(src, __builtin_amdgcn_readfirstlane(base + DY_BYTES + (wave + 4 * j) * 1024));
      }
    }
  };

  int tile = zsplit;
  int buf = 0;
  if (tile < a.ntiles) dma_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; tile < a.ntiles; tile += a.S) {
    if (tile + a.S < a.ntiles) dma_tile(tile + a.S, buf ^ 1);
    const char* const Ahi = smem + buf * BUF + wco * (NPX * 64) + frag_off;
    const char* const Bhi = smem + buf * BUF + DY_BYTES + wci * (XROWS * 64) + frag_off;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int m0 = ks * 16;
      const int py = m0 / TW, px = m0 - py * TW;
      const bf16x8 af = tr_frag(Ahi + m0 * 64, 4 * 64);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int hoff = ((py + t / 2) * HW_ + px + t % 2) * 64;
        const bf16x8 bfr = tr_frag(Bhi + hoff, 4 * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }

  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = co0 + wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      if (co < a.Cout && ci < cin)
        a.part[(((long)zsplit * NT + t) * a.Cout + co) * cin + ci] = acc[t][j];
    }
  }
}

// FLAT form of the S2D kernel above for the inner U-Net levels (maps of at most 8x8; see convflat_dma16_kernel): a
// pixel tile is 128 consecutive pixels of the FLATTENED batch, so sixteen 1x1 ... 4x4 images share one or two tiles
// instead of each occupying an eighth-full 8x16 window of its own (16 loop iterations of mostly zeros).  The halo
// image is replaced by one [128 pixels][32 ch] plane per tap, filled by per-lane DMA addresses.  Single-buffered: a
// workgroup sees one to eight tiles.  Same part[split][4][Cout][cin] slabs.
__global__ __launch_bounds__(256, 1) void conv2x2_wgrad_flat_kernel(WgradArgs a) {
  using T = bf16_t;
  constexpr int NT = 4, NPX = 128;
  constexpr int DY_BYTES = 2 * NPX * 64, PLANE = NPX * 64, X_BYTES = 2 * NT * PLANE;
  constexpr int DYG = 2 * NPX / 16 / 4;                  // 4
  constexpr int XG = 2 * NT * NPX / 16 / 4;              // 16
  constexpr int KSTEPS = NPX / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const T* __restrict__ dy = static_cast<const T*>(a.dy);
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wci = wave & 1;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int cin = a.c0;
  const int h = 1 << a.lgh, w = 1 << a.lgw;
  const int npix = a.B << (a.lgh + a.lgw);
  const int drow = lane >> 2, dslot = lane & 3;
  const int zsplit = (int)blockIdx.z;
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + p4 * 4) * 2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  auto dma_tile = [&](int tile) {
    const int p0 = tile * NPX;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int j = 0; j < DYG; ++j) {
      const int grp = wave + 4 * j;
      const int half = grp / (NPX / 16), m = (grp % (NPX / 16)) * 16 + drow;
      const int co = co0 + half * 32 + dslot * 8, p = p0 + m;
      const void* src = g_wgrad_zero_page;
      if (p < npix && co < a.Cout) src = dy + (long)p * a.lddy + co;
      dma16_asm(src, __builtin_amdgcn_readfirstlane(base + grp * 1024));
    }
#pragma unroll
    for (int j = 0; j < XG; ++j) {
      const int grp = wave + 4 * j;                          // 64 groups: [half][tap][8 groups of 16 pixels]
      const int half = grp >> 5, tap = (grp >> 3) & 3, m = (grp & 7) * 16 + drow;
      const int ci = ci0 + half * 32 + dslot * 8, p = p0 + m;
      const void* src = g_wgrad_zero_page;
      if (p < npix && ci < cin) {
        const int n = p >> (a.lgh + a.lgw), iy = (p >> a.lgw) & (h - 1), ix = p & (w - 1);
        const int rs = ci >> a.lgc, cc = ci & ((1 << a.lgc) - 1);
        const int py = 2 * (iy + (tap >> 1)) + (rs >> 1) - 1, px = 2 * (ix + (tap & 1)) + (rs & 1) - 1;
        if ((unsigned)py < (unsigned)(2 * h) && (unsigned)px < (unsigned)(2 * w))
          src = x0 + ((long)(n * 2 * h + py) * (2 * w) + px) * a.ld0 + cc;
      }
      dma16_asm(src, __builtin_amdgcn_readfirstlane(base + DY_BYTES + grp * 1024));
    }
  };

  for (int tile = zsplit; tile < a.ntiles; tile += a.S) {
    dma_tile(tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* const Ahi = smem + wco * (NPX * 64) + frag_off;
    const char* const Bhi = smem + DY_BYTES + wci * (NT * PLANE) + frag_off;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int m0 = ks * 16;
      const bf16x8 af = tr_frag(Ahi + m0 * 64, 4 * 64);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 bfr = tr_frag(Bhi + t * PLANE + m0 * 64, 4 * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // every wave has read this tile before the next one lands
  }
  const int r = lane & 31, hh = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = co0 + wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
      if (co < a.Cout && ci < cin) a.part[(((long)zsplit * NT + t) * a.Cout + co) * cin + ci] = acc[t][j];
    }
  }
}

// DIRECT form of the flat kernel for the same inner levels when the whole batch is at most 1024 pixels: no split over
// workgroups (the reduction is 16 ... 1024 pixels long), so there are no partial slabs and no fold launch -- the
// workgroup writes nn.Conv2d's [Cout][C][4][4] block itself.  For that it has to own all sixteen taps of its (o, c)
// pairs: its 64 virtual input channels are the four (r, s) sub-pixel groups of SIXTEEN real channels (the flat kernel
// takes 64 consecutive virtual channels = one (r, s) of 64 real ones, a quarter of every 64-byte [4][4] block), and the
// 64 x 16 x 16 tile leaves through LDS as 64 rows of 1 KiB.  Double-buffered over the (at most eight) pixel tiles.
__global__ __launch_bounds__(256, 2) void conv2x2_wgrad_small_kernel(WgradArgs a, float* __restrict__ grad, int accumulate) {
  using T = bf16_t;
  constexpr int NT = 4, NPX = 128;
  constexpr int DY_BYTES = 2 * NPX * 64, PLANE = NPX * 64, X_BYTES = 2 * NT * PLANE, BUF = DY_BYTES + X_BYTES;
  constexpr int DYG = 2 * NPX / 16 / 4;                  // 4
  constexpr int XG = 2 * NT * NPX / 16 / 4;              // 16
  constexpr int KSTEPS = NPX / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF]; the first 64 KiB again for the output tile
  const T* __restrict__ dy = static_cast<const T*>(a.dy);
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wci = wave & 1;
  const int c0 = blockIdx.x * 16, co0 = blockIdx.y * 64;
  const int C = a.c0 >> 2;                                // real input channels
  const int h = 1 << a.lgh, w = 1 << a.lgw;
  const int npix = a.B << (a.lgh + a.lgw);
  const int drow = lane >> 2, dslot = lane & 3;
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + p4 * 4) * 2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  auto dma_tile = [&](int tile, int buf) {
    const int p0 = tile * NPX;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + buf * BUF;
#pragma unroll
    for (int j = 0; j < DYG; ++j) {
      const int grp = wave + 4 * j;
      const int half = grp / (NPX / 16), m = (grp % (NPX / 16)) * 16 + drow;
      const int co = co0 + half * 32 + dslot * 8, p = p0 + m;
      const void* src = g_wgrad_zero_page;
      if (p < npix && co < a.Cout) src = dy + (long)p * a.lddy + co;
      dma16_asm(src, __builtin_amdgcn_readfirstlane(base + grp * 1024));
    }
#pragma unroll
    for (int j = 0; j < XG; ++j) {
      const int grp = wave + 4 * j;                          // 64 groups: [half][tap][8 groups of 16 pixels]
      const int half = grp >> 5, tap = (grp >> 3) & 3, m = (grp & 7) * 16 + drow;
      const int vci = half * 32 + dslot * 8, rs = vci >> 4, cc = c0 + (vci & 15), p = p0 + m;
      const void* src = g_wgrad_zero_page;
      if (p < npix && cc < C) {
        const int n = p >> (a.lgh + a.lgw), iy = (p >> a.lgw) & (h - 1), ix = p & (w - 1);
        const int py = 2 * (iy + (tap >> 1)) + (rs >> 1) - 1, px = 2 * (ix + (tap & 1)) + (rs & 1) - 1;
        if ((unsigned)py < (unsigned)(2 * h) && (unsigned)px < (unsigned)(2 * w))
          src = x0 + ((long)(n * 2 * h + py) * (2 * w) + px) * a.ld0 + cc;
      }
      dma16_asm(src, __builtin_amdgcn_readfirstlane(base + DY_BYTES + grp * 1024));
    }
  };

  int buf = 0;
  dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int tile = 0; tile < a.ntiles; ++tile) {
    if (tile + 1 < a.ntiles) dma_tile(tile + 1, buf ^ 1);
    const char* const Ahi = smem + buf * BUF + wco * (NPX * 64) + frag_off;
    const char* const Bhi = smem + buf * BUF + DY_BYTES + wci * (NT * PLANE) + frag_off;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int m0 = ks * 16;
      const bf16x8 af = tr_frag(Ahi + m0 * 64, 4 * 64);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 bfr = tr_frag(Bhi + t * PLANE + m0 * 64, 4 * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // the next tile has landed and every wave is done with this one
    buf ^= 1;
  }
  // [64 o][16 c][4][4] through LDS: lane = (virtual channel r -> (rs, c), half hh of the co pattern)
  float* const stg = reinterpret_cast<float*>(smem);
  const int r = lane & 31, hh = lane >> 5;
  const int rs = wci * 2 + (r >> 4), c = r & 15;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int t16 = (2 * (t >> 1) + (rs >> 1)) * 4 + 2 * (t & 1) + (rs & 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int col = wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
      stg[(col * 16 + c) * 16 + t16] = acc[t][j];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int pi = tid + k * 256, col = pi >> 6, qq = pi & 63;
    if (co0 + col < a.Cout) {
      float* const dst = grad + ((long)(co0 + col) * C + c0) * 16 + qq * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + col * 256 + qq * 4);
      if (accumulate) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += o[i];
      }
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  }
}

// KS x KS taps, stride 1 (row a13: the PatchGAN's 4x4 stride-1 pad-1 layers, KS = 4, PAD = 1):
//   dW[kh*KS+kw][co][ci] = sum_{n,i,j} dY[n][i][j][co] * X[n][i+kh-PAD][j+kw-PAD][ci],   X is (H+KS-1-2PAD) wide.
// Sixteen 32x32 accumulators do not fit a wave, so the kernel rows go to KS workgroups (blockIdx.z = split*KS + kh,
// the NT = 3 idea of the 3x3 kernel): each accumulates the KS taps of its row and stages only the TH halo rows that
// row reads.  Every lane bounds-tests its halo pixel (the padded border is part of most tiles here).
template <int TH, int TW, int KS, int PAD>
__global__ __launch_bounds__(256, 2) void convkxk_wgrad_rows_kernel(WgradArgs a) {
  using T = bf16_t;
  constexpr int NT = KS;
  constexpr int NPX = TH * TW;
  constexpr int HW_ = TW + KS - 1, HR = TH, HALO = HR * HW_;
  constexpr int XROWS = (HALO + 31) / 32 * 32;
  constexpr int DY_BYTES = 2 * NPX * 64, X_BYTES = 2 * XROWS * 64;
  constexpr int DYG = 2 * NPX / 16 / 4;
  constexpr int XG = 2 * XROWS / 16 / 4;
  constexpr int BUF = DY_BYTES + X_BYTES;
  constexpr int KSTEPS = NPX / 16;
  static_assert((2 * NPX / 16) % 4 == 0 && (2 * XROWS / 16) % 4 == 0, "DMA groups split evenly over 4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF]
  const T* __restrict__ dy = static_cast<const T*>(a.dy);
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wci = wave & 1;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int cin = a.c0;
  const int Hi = a.H + KS - 1 - 2 * PAD, Wi = a.W + KS - 1 - 2 * PAD;
  const int drow = lane >> 2, dslot = lane & 3;
  const int kh = (int)(blockIdx.z % KS), zsplit = (int)(blockIdx.z / KS);

  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + p * 4) * 2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  int dy_off[DYG], dy_py[DYG], dy_px[DYG];
  bool dy_ok[DYG];
#pragma unroll
  for (int j = 0; j < DYG; ++j) {
    const int grp = wave + 4 * j;
    const int half = grp / (NPX / 16), px = (grp % (NPX / 16)) * 16 + drow;
    const int co = co0 + half * 32 + dslot * 8;
    dy_py[j] = px / TW; dy_px[j] = px - dy_py[j] * TW;
    dy_ok[j] = co < a.Cout;
    dy_off[j] = (dy_py[j] * a.W + dy_px[j]) * a.lddy + co;
  }
  int x_ch[XG], x_hy[XG], x_hx[XG];
#pragma unroll
  for (int j = 0; j < XG; ++j) {
    const int grp = wave + 4 * j;
    const int half = grp / (XROWS / 16), px = (grp % (XROWS / 16)) * 16 + drow;
    const int ci = ci0 + half * 32 + dslot * 8;
    x_hy[j] = px / HW_; x_hx[j] = px - x_hy[j] * HW_;
    x_ch[j] = (px < HALO && ci < cin) ? ci : -1;
  }

  auto dma_tile = [&](int tile, int buf) {
    int bt = tile;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int img = bt / a.tilesY;
    const int y0 = ty * TH, xs = tx * TW;
    const long pixbase = (long)(img * a.H + y0) * a.W + xs;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + buf * BUF;
#pragma unroll
    for (int j = 0; j < DYG; ++j) {
      const int gy = y0 + dy_py[j], gx = xs + dy_px[j];
      const void* src = g_wgrad_zero_page;
      if (gy < a.H && gx < a.W && dy_ok[j]) src = dy + pixbase * a.lddy + dy_off[j];
      dma16_asm(src, __builtin_amdgcn_readfirstlane(base + (wave + 4 * j) * 1024));
    }
#pragma unroll
    for (int j = 0; j < XG; ++j) {
      const int gy = y0 + kh - PAD + x_hy[j], gx = xs - PAD + x_hx[j];
      const void* src = g_wgrad_zero_page;
      if (x_ch[j] >= 0 && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
        src = x0 + ((long)(img * Hi + gy) * Wi + gx) * a.ld0 + x_ch[j];
      dma16_asm(src, __builtin_amdgcn_readfirstlane(base + DY_BYTES + (wave + 4 * j) * 1024));
    }
  };

  int tile = zsplit;
  int buf = 0;
  if (tile < a.ntiles) dma_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; tile < a.ntiles; tile += a.S) {
    if (tile + a.S < a.ntiles) dma_tile(tile + a.S, buf ^ 1);
    const char* const Ahi = smem + buf * BUF + wco * (NPX * 64) + frag_off;
    const char* const Bhi = smem + buf * BUF + DY_BYTES + wci * (XROWS * 64) + frag_off;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int m0 = ks * 16;
      const int py = m0 / TW, px = m0 - py * TW;
      const bf16x8 af = tr_frag(Ahi + m0 * 64, 4 * 64);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int hoff = (py * HW_ + px + t) * 64;
        const bf16x8 bfr = tr_frag(Bhi + hoff, 4 * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }

  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = kh * KS + t;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = co0 + wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      if (co < a.Cout && ci < cin)
        a.part[(((long)zsplit * (KS * KS) + tap) * a.Cout + co) * cin + ci] = acc[t][j];
    }
  }
}

// Register-staged form of the kernel above for either dtype (T = float: three-way bf16 split of both operands, the
// parity mode; see conv3x3_wgrad_kernel).  Same row split over workgroups (blockIdx.z = split*KS + kh), same
// part[split][KS*KS][Cout][cin] slabs; any KS / PAD of row a13 (2x2 pad 0: the 4x4 stride-2 layers on the
// space-to-depth image; 4x4 pad 1: the PatchGAN's stride-1 layers).
template <typename T, int TH, int TW, int KS, int PAD>
__global__ __launch_bounds__(256, (std::is_same<T, float>::value ? 1 : 2)) void convkxk_wgrad_rs_kernel(WgradArgs a) {
  constexpr bool SPLIT = std::is_same<T, float>::value;
  constexpr int NIMG = SPLIT ? 3 : 1;
  constexpr int NT = KS;
  constexpr int NPX = TH * TW;
  constexpr int HW_ = TW + KS - 1, HALO = TH * HW_;
  constexpr int DY_BYTES = 2 * NPX * 64;
  constexpr int X_BYTES = 2 * HALO * 64;
  constexpr int DY_PIECES = NPX * 8, DY_IT = (DY_PIECES + 255) / 256;
  constexpr int X_PIECES = HALO * 8, X_IT = (X_PIECES + 255) / 256;
  constexpr int KSTEPS = NPX / 16;
  static_assert(TW % 16 == 0, "a k-step is 16 consecutive pixels of one tile row");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsDY = smem;
  char* const ldsX = smem + NIMG * DY_BYTES;
  const T* __restrict__ dy = static_cast<const T*>(a.dy);
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave >> 1, wci = wave & 1;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int cin = a.c0;
  const int Hi = a.H + KS - 1 - 2 * PAD, Wi = a.W + KS - 1 - 2 * PAD;
  const int kh = (int)(blockIdx.z % KS), zsplit = (int)(blockIdx.z / KS);
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int frag_off = ((g >> 1) * 8 + q) * 64 + ((g & 1) * 16 + p * 4) * 2;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  WPiece<T> dreg[DY_IT];
  WPiece<T> xreg[X_IT];

  auto load_tile = [&](int tile) {
    int bt = tile;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    const int img = bt / a.tilesY;
    const int y0 = ty * TH, xs = tx * TW;
#pragma unroll
    for (int i = 0; i < DY_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      const int py = px / TW, pxx = px - py * TW;
      const int gy = y0 + py, gx = xs + pxx, co = co0 + pc * 8;
      dreg[i].zero();
      if (idx < DY_PIECES && gy < a.H && gx < a.W && co < a.Cout)
        dreg[i].load(dy + ((long)(img * a.H + gy) * a.W + gx) * a.lddy + co);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      const int hy = px / HW_, hx = px - hy * HW_;
      const int gy = y0 + kh - PAD + hy, gx = xs - PAD + hx, ci = ci0 + pc * 8;
      xreg[i].zero();
      if (idx < X_PIECES && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi && ci < cin)
        xreg[i].load(x0 + ((long)(img * Hi + gy) * Wi + gx) * a.ld0 + ci);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < DY_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      if (idx < DY_PIECES) dreg[i].to_lds(ldsDY, DY_BYTES, (pc >> 2) * (NPX * 64) + px * 64 + (pc & 3) * 16);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int idx = tid + i * 256;
      const int px = idx >> 3, pc = idx & 7;
      if (idx < X_PIECES) xreg[i].to_lds(ldsX, X_BYTES, (pc >> 2) * (HALO * 64) + px * 64 + (pc & 3) * 16);
    }
  };
  const char* const Ahi = ldsDY + wco * (NPX * 64) + frag_off;
  const char* const Bhi = ldsX + wci * (HALO * 64) + frag_off;

  int tile = zsplit;
  if (tile < a.ntiles) load_tile(tile);
  for (; tile < a.ntiles; tile += a.S) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + a.S < a.ntiles) load_tile(tile + a.S);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int m0 = ks * 16;
      const int py = m0 / TW, px = m0 - py * TW;
      const bf16x8 af = tr_frag(Ahi + m0 * 64, 4 * 64);
      bf16x8 am, al;
      if constexpr (SPLIT) {
        am = tr_frag(Ahi + DY_BYTES + m0 * 64, 4 * 64);
        al = tr_frag(Ahi + 2 * DY_BYTES + m0 * 64, 4 * 64);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int hoff = (py * HW_ + px + t) * 64;
        const bf16x8 bfr = tr_frag(Bhi + hoff, 4 * 64);
        if constexpr (SPLIT) {
          const bf16x8 bm = tr_frag(Bhi + X_BYTES + hoff, 4 * 64);
          const bf16x8 bl = tr_frag(Bhi + 2 * X_BYTES + hoff, 4 * 64);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bfr, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bfr, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bm, acc[t], 0, 0, 0);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
      }
    }
  }
  const int r = lane & 31, h = lane >> 5;
  const int ci = ci0 + wci * 32 + r;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = kh * KS + t;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = co0 + wco * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      if (co < a.Cout && ci < cin) a.part[(((long)zsplit * (KS * KS) + tap) * a.Cout + co) * cin + ci] = acc[t][j];
    }
  }
}

// Fold the split slabs of a 4x4 layer straight into nn.Conv2d's [Cout][C][4][4] gradient (what the optimiser's flat
// buffer holds), in a fixed order:
//   mode 1 (stride-2 layers, slabs [4 taps (a,b)][Cout][4*C] over the space-to-depth channels (r,s,c)):
//           grad[o][c][2a+r][2b+s] (+)= sum_z part[z][a*2+b][o][(r*2+s)*C + c]
//   mode 2 (stride-1 layers, slabs [16 taps (kh,kw)][Cout][C]):  grad[o][c][kh][kw] (+)= sum_z part[z][kh*4+kw][o][c]
// A workgroup owns one o and 64 consecutive c: its four waves read the sixteen 256-byte rows of every slab, the sums
// meet in LDS and leave as 1024 contiguous floats.
template <int ZG>
__global__ __launch_bounds__(256 * ZG) void wgrad_fold4x4_kernel(const float* __restrict__ part, float* __restrict__ grad, int S,
                                                                 int Cout, int C, int mode, int accumulate) {
  // ZG groups of four waves take every ZG-th slab (a layer with few (o, c) tiles has up to 256 slabs to fold, so the
  // parallelism has to come from inside the workgroup); eight loads in flight per thread
  __shared__ float tile[ZG][16][65];
  const int o = blockIdx.y, c0 = blockIdx.x * 64;
  const int cl = threadIdx.x & 63, grp = (threadIdx.x >> 6) & 3, zg = threadIdx.x >> 8;
  const long slab = 16L * Cout * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t16 = grp + 4 * i, kh = t16 >> 2, kw = t16 & 3;
    long src;
    if (mode == 1) src = ((long)((kh >> 1) * 2 + (kw >> 1)) * Cout + o) * (4L * C) + ((kh & 1) * 2 + (kw & 1)) * C + c0 + cl;
    else src = ((long)t16 * Cout + o) * C + c0 + cl;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (c0 + cl < C) {
      int z = zg;
      for (; z + 7 * ZG < S; z += 8 * ZG) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += part[(long)(z + k * ZG) * slab + src];
      }
      for (; z < S; z += ZG) acc[0] += part[(long)z * slab + src];
    }
    tile[zg][t16][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  __syncthreads();
  const long base = ((long)o * C + c0) * 16;
  for (int k = threadIdx.x; k < 1024; k += 256 * ZG) {
    const int c = k >> 4, t16 = k & 15;
    if (c0 + c < C) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < ZG; ++g) v += tile[g][t16][c];
      grad[base + k] = accumulate ? grad[base + k] + v : v;
    }
  }
}

// out[i] (+)= sum_z part[z][i], i < n (n = 4 * Cout * K): the split slabs of the kernel above, folded in a fixed order
__global__ __launch_bounds__(256) void split_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int S, long n,
                                                        int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f;
  int z = 0;
  for (; z + 1 < S; z += 2) { s0 += part[(long)z * n + i]; s1 += part[(long)(z + 1) * n + i]; }
  if (z < S) s0 += part[(long)z * n + i];
  const float v = s0 + s1;
  out[i] = accumulate ? out[i] + v : v;
}

// workgroups the 4x4 weight-gradient launches aim at (S2S_P2P_WGRAD_BLOCKS; see s2s_conv3x3_wgrad_splits)
int p2p_wgrad_target() {
  static const int t = [] { const char* e = getenv("S2S_P2P_WGRAD_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();   // (512: 4.07, 384: 4.04, 256: 4.01, 192: 4.28 ms per G + D step on one box)
  return t;
}

int conv2x2_wgrad_splits(int B, int H, int W, int Cin, int Cout) {
  const int nt = B * cdiv(H, 8) * cdiv(W, 16);
  const int mn = cdiv(Cin, 64) * cdiv(Cout, 64);
  int s = p2p_wgrad_target() / mn;
  if (s > 256) s = 256;
  if (s > nt) s = nt;
  if (s < 1) s = 1;
  return s;
}

// one pixel tile = 8 rows x 16 columns (halo 10 x 18); this shape keeps the staging prefetch small
// enough for two workgroups per CU (a 4 x 32 tile spills at that occupancy)
int wgrad_ntiles(int B, int H, int W) { return B * cdiv(H, 8) * cdiv(W, 16); }

}  // namespace

// Split count the kernel will use for this problem; the caller sizes `part` as
// [splits][9][Cout][Cin] floats.
extern "C" int s2s_conv3x3_wgrad_splits(int dtype, int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return S2S_ERR_SHAPE;
  const int nt = wgrad_ntiles(B, H, W);
  const int mn = cdiv(Cin, 64) * cdiv(Cout, 64);
  static const int target_env = [] { const char* e = getenv("S2S_WGRAD_BLOCKS"); return e ? atoi(e) : 0; }();
  // ONE workgroup per CU (256; round 3).  Alone, the kernel is 7 % faster with two per CU (1078 against 1003 TFLOP/s), but
  // a pair takes every vector register of its CU, and this kernel runs on the side stream UNDER the bandwidth-bound
  // BatchNorm / up-sampling backward passes of the compute stream: with half of each CU's registers left free those
  // passes are really co-resident instead of waiting for a CU to drain -- CFM step 7.75 -> 7.50 ms on one box (-3.3 %),
  // and the split slabs (and their fold: 0.26 -> 0.17 ms per step) halve.  Sweep on one box (ms per step): 512: 7.72-7.77,
  // 384: 7.58-7.64, 256: 7.45-7.52, 224: 7.49, 192: 7.59, 128: 7.93.  S2S_WGRAD_BLOCKS=512 restores two per CU.
  const int target = target_env ? target_env : 256;   // (384 / 512 for the concat-input layers alone: 7.48 / 7.52 against 7.45 ms)
  // every split costs a |dW| x 4 B partial slab; the cap only matters for targets above 320
  const int cap = mn == 1 ? 512 : 320;   // 64->64, 256x256, batch 16, with the 16-wave reduce: 400 -> 110 us, 512 -> 103 us
  int s = target / mn;
  // (a layer whose (co, ci) tiles alone are 3/4 of the target -- 1536 -> 512: 192 tiles -- would leave a quarter of the CUs
  //  without a workgroup: one more split)
  if (target_env == 0 && s >= 1 && (long)s * mn < (long)target * 7 / 8 && (long)(s + 1) * mn <= 2L * target) s += 1;
  if (s > cap) s = cap;
  if (s > nt) s = nt;
  if (s < 1) s = 1;
  return s;
}

// phases: 1 = the split MFMA kernel (partial slabs into `part`), 2 = the fold of the slabs into the OIHW gradient,
// 3 = both.  Two calls with phases 1 and 2 equal one call with 3; a profiler uses them to time the two kernels apart.
extern "C" int s2s_conv3x3_wgrad_phase(int dtype, const void* dy, int lddy, int Cout, const void* x0, int ld0, int c0,
                                       const void* x1, int ld1, int c1, float* part, float* grad_oihw,
                                       int accumulate, int B, int H, int W, int phases, void* stream) {
  if (!dy || !x0 || !part || !grad_oihw) return S2S_ERR_NULL;
  if (phases < 1 || phases > 3) return S2S_ERR_SHAPE;
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || c0 <= 0 || c1 < 0) return S2S_ERR_SHAPE;
  if ((Cout % 8) || (c0 % 8) || (c1 % 8) || (lddy % 8) || (ld0 % 8) || (ld1 % 8) || (c1 > 0 && !x1)) return S2S_ERR_SHAPE;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x0 & 15) || ((uintptr_t)x1 & 15)) return S2S_ERR_ALIGN;
  WgradArgs a{};
  a.dy = dy; a.x0 = x0; a.x1 = x1; a.part = part;
  a.lddy = lddy; a.Cout = Cout; a.ld0 = ld0; a.c0 = c0; a.ld1 = ld1; a.c1 = c1;
  a.B = B; a.H = H; a.W = W; a.lgc = 0; a.lgh = a.lgw = 0;
  a.S = s2s_conv3x3_wgrad_splits(dtype, B, H, W, c0 + c1, Cout);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = S2S_OK;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  if (!(phases & 1)) { if (a.S > wgrad_ntiles(B, H, W)) return S2S_ERR_SHAPE; }
  else if (dtype == S2S_BF16) rc = launch_wgrad_dma<8, 16>(a, s);
  else rc = launch_wgrad<float, 8, 16>(a, s);
  if (rc != S2S_OK || !(phases & 2)) return rc;
  const long n = (long)Cout * (c0 + c1);
  if (n <= 32768 && a.S >= 32)
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s, part, grad_oihw,
                       a.S, Cout, c0 + c1, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, part, grad_oihw,
                       a.S, Cout, c0 + c1, accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_conv3x3_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x0, int ld0, int c0,
                                      const void* x1, int ld1, int c1, float* part, float* grad_oihw,
                                      int accumulate, int B, int H, int W, void* stream) {
  return s2s_conv3x3_wgrad_phase(dtype, dy, lddy, Cout, x0, ld0, c0, x1, ld1, c1, part, grad_oihw, accumulate, B, H, W, 3,
                                 stream);
}

// ---- 2x2-tap weight gradient (row a13) --------------------------------------------------------------------------
// grad2[tap a*2+b][Cout][cin] (+)= sum over the batch; dY is [B][H][W][Cout], X is [B][H+1][W+1][cin] (the pad = 0
// operand of s2s_conv2x2_nhwc).  part: float[s2s_conv2x2_wgrad_splits()][4][Cout][cin].
extern "C" int s2s_conv2x2_wgrad_splits(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return S2S_ERR_SHAPE;
  return conv2x2_wgrad_splits(B, H, W, Cin, Cout);
}

extern "C" int s2s_conv2x2_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x, int ldx, int cin,
                                      float* part, float* grad2, int accumulate, int B, int H, int W, void* stream) {
  if (!dy || !x || !part || !grad2) return S2S_ERR_NULL;
  if (dtype != S2S_BF16) return S2S_ERR_DTYPE;
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || cin <= 0) return S2S_ERR_SHAPE;
  if ((Cout % 8) || (cin % 8) || (lddy % 8) || (ldx % 8)) return S2S_ERR_SHAPE;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return S2S_ERR_ALIGN;
  if ((long)B * (H + 1) * (W + 1) >= (1L << 31)) return S2S_ERR_SHAPE;
  WgradArgs a{};
  a.dy = dy; a.x0 = x; a.x1 = nullptr; a.part = part;
  a.lddy = lddy; a.Cout = Cout; a.ld0 = ldx; a.c0 = cin; a.ld1 = 8; a.c1 = 0;
  a.B = B; a.H = H; a.W = W; a.lgc = 0; a.lgh = a.lgw = 0;
  a.S = conv2x2_wgrad_splits(B, H, W, cin, Cout);
  a.tilesY = cdiv(H, 8); a.tilesX = cdiv(W, 16);
  a.ntiles = B * a.tilesY * a.tilesX;
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int TH = 8, TW = 16;
  constexpr int XROWS = ((TH + 1) * (TW + 1) + 31) / 32 * 32;
  constexpr int lds = 2 * (2 * TH * TW * 64 + 2 * XROWS * 64);
  auto kern = conv2x2_wgrad_dma_kernel<TH, TW>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(cdiv(cin, 64), cdiv(Cout, 64), a.S);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  const long n = 4L * Cout * cin;
  hipLaunchKernelGGL(split_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, grad2, a.S, n,
                     accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// ---- 4x4 stride-1 weight gradient (row a13, PatchGAN layers) ------------------------------------------------------
// grad16[tap kh*4+kw][Cout][cin] (+)= sum over the batch; dY is [B][H][W][Cout], X is [B][H+1][W+1][cin] (pad 1).
// part: float[s2s_conv4x4s1_wgrad_splits()][16][Cout][cin].
extern "C" int s2s_conv4x4s1_wgrad_splits(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return S2S_ERR_SHAPE;
  const int nt = B * cdiv(H, 8) * cdiv(W, 16);
  const int mn = cdiv(Cin, 64) * cdiv(Cout, 64) * 4;        // four kernel rows per (co, ci) tile
  // (the stride-1 PatchGAN layers -- C512 is 129 GFLOP, a fifth of the step's weight-gradient work -- do better at 1.5
  //  workgroups per CU: 3.82 against 3.85 ms per G + D step; S2S_P2P_WGRAD_S1_BLOCKS)
  static const int t1 = [] { const char* e = getenv("S2S_P2P_WGRAD_S1_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 384; }();
  int s = t1 / mn;
  if (s > 128) s = 128;
  if (s > nt) s = nt;
  if (s < 1) s = 1;
  return s;
}

extern "C" int s2s_conv4x4s1_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x, int ldx, int cin,
                                        float* part, float* grad16, int accumulate, int B, int H, int W,
                                        void* stream) {
  if (!dy || !x || !part || !grad16) return S2S_ERR_NULL;
  if (dtype != S2S_BF16) return S2S_ERR_DTYPE;
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || cin <= 0) return S2S_ERR_SHAPE;
  if ((Cout % 8) || (cin % 8) || (lddy % 8) || (ldx % 8)) return S2S_ERR_SHAPE;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return S2S_ERR_ALIGN;
  if ((long)B * (H + 1) * (W + 1) >= (1L << 31)) return S2S_ERR_SHAPE;
  WgradArgs a{};
  a.dy = dy; a.x0 = x; a.x1 = nullptr; a.part = part;
  a.lddy = lddy; a.Cout = Cout; a.ld0 = ldx; a.c0 = cin; a.ld1 = 8; a.c1 = 0;
  a.B = B; a.H = H; a.W = W; a.lgc = 0; a.lgh = a.lgw = 0;
  a.S = s2s_conv4x4s1_wgrad_splits(B, H, W, cin, Cout);
  a.tilesY = cdiv(H, 8); a.tilesX = cdiv(W, 16);
  a.ntiles = B * a.tilesY * a.tilesX;
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int TH = 8, TW = 16, KS = 4;
  constexpr int XROWS = (TH * (TW + KS - 1) + 31) / 32 * 32;
  constexpr int lds = 2 * (2 * TH * TW * 64 + 2 * XROWS * 64);
  auto kern = convkxk_wgrad_rows_kernel<TH, TW, KS, 1>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(cdiv(cin, 64), cdiv(Cout, 64), a.S * KS);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  const long n = 16L * Cout * cin;
  hipLaunchKernelGGL(split_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, grad16, a.S, n,
                     accumulate);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// ---- general weight gradient of the row-a13 layers ----------------------------------------------------------------
// ks = 2: dY [B][H][W][Cout], X [B][H+1][W+1][cin] (the pad-0 operand of s2s_convkxk_nhwc: the space-to-depth image,
//         cin = 4 C); ks = 4: dY [B][H][W][Cout], X [B][H+1][W+1][cin] (pad 1, cin = C).
// layout 0: grad[ks*ks][Cout][cin] (the kernel's own tap-major form); layout 1: nn.Conv2d's [Cout][C][4][4] (for the
// transposed layers call it with the roles of the layer's input and output gradient exchanged: the result is
// nn.ConvTranspose2d's [Cin][Cout][4][4]).  dtype fp32 = parity mode (register-staged, three-way split).
// part: float[s2s_convkxk_wgrad_splits()][ks*ks][Cout][cin].
extern "C" int s2s_convkxk_wgrad_splits(int dtype, int B, int H, int W, int Cin, int Cout, int ks) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ks != 2 && ks != 4)) return S2S_ERR_SHAPE;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  if (dtype == S2S_BF16 && ks == 2) return conv2x2_wgrad_splits(B, H, W, Cin, Cout);
  if (dtype == S2S_BF16) return s2s_conv4x4s1_wgrad_splits(B, H, W, Cin, Cout);
  const int nt = B * cdiv(H, 8) * cdiv(W, 16);
  const int mn = cdiv(Cin, 64) * cdiv(Cout, 64) * ks;
  int s = 512 / mn;
  if (s > 128) s = 128;
  if (s > nt) s = nt;
  if (s < 1) s = 1;
  return s;
}

template <typename T, int KS, int PAD>
static int launch_wgrad_rs(WgradArgs& a, hipStream_t s) {
  constexpr int TH = 8, TW = 16;
  constexpr int NIMG = std::is_same<T, float>::value ? 3 : 1;
  constexpr int lds = NIMG * (2 * TH * TW * 64 + 2 * TH * (TW + KS - 1) * 64);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = convkxk_wgrad_rs_kernel<T, TH, TW, KS, PAD>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(cdiv(a.c0, 64), cdiv(a.Cout, 64), a.S * KS);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  return S2S_OK;
}

extern "C" int s2s_convkxk_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x, int ldx, int cin,
                                      float* part, float* grad, int layout, int accumulate, int B, int H, int W, int ks,
                                      int x_plain, void* stream) {
  if (!dy || !x || !part || !grad) return S2S_ERR_NULL;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || cin <= 0 || (ks != 2 && ks != 4)) return S2S_ERR_SHAPE;
  if ((Cout % 8) || (cin % 8) || (lddy % 8) || (ldx % 8) || (layout != 0 && layout != 1)) return S2S_ERR_SHAPE;
  if (ks == 2 && layout == 1 && (cin % 32)) return S2S_ERR_SHAPE;          // cin = 4 C with C a multiple of 8
  // x_plain: X is the plain [B][2H][2W][C] tensor (bf16, ks = 2, C a power of two), space-to-depth done by the loader
  if (x_plain && (dtype != S2S_BF16 || ks != 2 || (cin % 32) || ((cin / 4) & (cin / 4 - 1)) || ldx < cin / 4)) return S2S_ERR_SHAPE;
  if (x_plain && (H >= 16384 || W >= 16384)) return S2S_ERR_SHAPE;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return S2S_ERR_ALIGN;
  if ((long)B * (H + 1) * (W + 1) >= (1L << 31)) return S2S_ERR_SHAPE;
  WgradArgs a{};
  a.dy = dy; a.x0 = x; a.x1 = nullptr; a.part = part;
  a.lddy = lddy; a.Cout = Cout; a.ld0 = ldx; a.c0 = cin; a.ld1 = 8; a.c1 = 0;
  a.B = B; a.H = H; a.W = W; a.lgc = 0; a.lgh = a.lgw = 0;
  a.S = s2s_convkxk_wgrad_splits(dtype, B, H, W, cin, Cout, ks);
  a.tilesY = cdiv(H, 8); a.tilesX = cdiv(W, 16);
  a.ntiles = B * a.tilesY * a.tilesX;
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int TH = 8, TW = 16;
  int rc = S2S_OK;
  if (dtype == S2S_F32) {
    rc = ks == 2 ? launch_wgrad_rs<float, 2, 0>(a, s) : launch_wgrad_rs<float, 4, 1>(a, s);
  } else if (ks == 2) {
    constexpr int XROWS = ((TH + 1) * (TW + 1) + 31) / 32 * 32;
    constexpr int lds = 2 * (2 * TH * TW * 64 + 2 * XROWS * 64);
    if (x_plain && H <= 8 && W <= 8 && (H & (H - 1)) == 0 && (W & (W - 1)) == 0) {
      // inner U-Net levels: tiles of the flattened batch (fewer splits than the slab buffer was sized for)
      a.lgc = __builtin_ctz((unsigned)(cin / 4));
      a.lgh = __builtin_ctz((unsigned)H); a.lgw = __builtin_ctz((unsigned)W);
      a.ntiles = cdiv(B * H * W, 128);
      if (layout == 1 && a.ntiles <= 8 && ((cin / 4) % 16) == 0) {
        // at most 1024 pixels in the whole batch: one workgroup per (64 o, 16 c) block reduces all of them and writes the
        // [Cout][C][4][4] gradient itself -- no slabs, no fold launch
        constexpr int lds_buf = 2 * 128 * 64 + 2 * 4 * 128 * 64;
        static unsigned long long attr_small = 0;   // hipFuncSetAttribute is per device
        if (int rc2 = s2s_allow_dyn_lds(reinterpret_cast<const void*>(conv2x2_wgrad_small_kernel), 2 * lds_buf, &attr_small)) return rc2;
        // (a single pixel tile needs one buffer: 80 KiB, so two workgroups share a CU and one's stores run under the other's loads)
        hipLaunchKernelGGL(conv2x2_wgrad_small_kernel, dim3(cin / 64, cdiv(Cout, 64)), dim3(256), (a.ntiles == 1 ? 1 : 2) * lds_buf, s, a, grad, accumulate);
        S2S_LAUNCH_CHECK();
        return S2S_OK;
      }
      const int mn = cdiv(cin, 64) * cdiv(Cout, 64);
      int sp = p2p_wgrad_target() / mn;
      if (sp > a.ntiles) sp = a.ntiles;
      if (sp > a.S) sp = a.S;
      a.S = sp < 1 ? 1 : sp;
      constexpr int lds_flat = 2 * 128 * 64 + 2 * 4 * 128 * 64;
      static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
      if (int rc2 = s2s_allow_dyn_lds(reinterpret_cast<const void*>(conv2x2_wgrad_flat_kernel), lds_flat, &attr_devs)) return rc2;
      hipLaunchKernelGGL(conv2x2_wgrad_flat_kernel, dim3(cdiv(cin, 64), cdiv(Cout, 64), a.S), dim3(256), lds_flat, s, a);
    } else if (x_plain) {
      a.lgc = __builtin_ctz((unsigned)(cin / 4));
      auto kern = conv2x2_wgrad_dma_kernel<TH, TW, true>;
      static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
      if (int rc2 = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc2;
      hipLaunchKernelGGL(kern, dim3(cdiv(cin, 64), cdiv(Cout, 64), a.S), dim3(256), lds, s, a);
    } else {
      auto kern = conv2x2_wgrad_dma_kernel<TH, TW, false>;
      static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
      if (int rc2 = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc2;
      hipLaunchKernelGGL(kern, dim3(cdiv(cin, 64), cdiv(Cout, 64), a.S), dim3(256), lds, s, a);
    }
  } else {
    constexpr int KS = 4;
    constexpr int XROWS = (TH * (TW + KS - 1) + 31) / 32 * 32;
    constexpr int lds = 2 * (2 * TH * TW * 64 + 2 * XROWS * 64);
    auto kern = convkxk_wgrad_rows_kernel<TH, TW, KS, 1>;
    static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
    if (int rc2 = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc2;
    hipLaunchKernelGGL(kern, dim3(cdiv(cin, 64), cdiv(Cout, 64), a.S * KS), dim3(256), lds, s, a);
  }
  if (rc != S2S_OK) return rc;
  if (layout == 0) {
    const long n = (long)ks * ks * Cout * cin;
    hipLaunchKernelGGL(split_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, grad, a.S, n, accumulate);
  } else {
    const int C = ks == 2 ? cin / 4 : cin;
    if (a.S >= 16 && (long)cdiv(C, 64) * Cout <= 1024)
      hipLaunchKernelGGL(wgrad_fold4x4_kernel<4>, dim3(cdiv(C, 64), Cout), dim3(1024), 0, s, part, grad, a.S, Cout, C,
                         ks == 2 ? 1 : 2, accumulate);
    else
      hipLaunchKernelGGL(wgrad_fold4x4_kernel<1>, dim3(cdiv(C, 64), Cout), dim3(256), 0, s, part, grad, a.S, Cout, C,
                         ks == 2 ? 1 : 2, accumulate);
  }
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
