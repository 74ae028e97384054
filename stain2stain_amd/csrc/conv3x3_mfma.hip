// 3x3 / stride 1 / pad 1 convolution, NHWC, im2col-free implicit GEMM on MFMA (gfx950).
//
// Replaces the ATen conv2d behind nn.Conv2d(k=3, padding=1) of the reference's DoubleConv
// (src/models/components/shared_encoder.py:15,18; task_decoders.py:15,18) in the forward pass, and
// the data-gradient of the same layers in the backward pass (same kernel, weights packed
// flipped/transposed by s2s_pack_conv3x3).
//
// Decomposition: D[pixel][cout] = sum_{tap, cin} X[pixel (+) tap][cin] * Wp[cin/32][tap][cout][32]
//   * one workgroup (256 threads = 4 waves) owns a TH x TW spatial tile of one image and BN output
//     channels; the accumulators (fp32) live in registers for the whole K loop.
//   * K loop: input channels in chunks of 32.  Per chunk the (TH+2) x (TW+2) x 32 halo patch is
//     staged ONCE into LDS and re-read by all nine taps (this is what replaces im2col: the tap is
//     a constant LDS row offset), and the nine [BN][32] weight slabs stream through a two-slot
//     LDS ring, one barrier per tap; global loads for the next slab / next halo are in flight
//     (registers) while the current tap's MFMAs run.
//   * LDS rows are 32 bf16 + 16 B pad = 80 B, which makes every ds_read_b128 fragment read
//     conflict-free (16 consecutive rows land on 16 distinct 16-B slots of the 256-B bank row).
//   * v_mfma_f32_32x32x16_bf16, A = pixel fragment, B = cout fragment, so each lane ends up with
//     one output channel (column) and 16 pixels (rows): per-channel BatchNorm partial sums are
//     lane-local and cost one cross-half shuffle.
//   * the input can come from two tensors (channels [0,c0) from x0, [c0,c0+c1) from x1): this is the
//     decoder's torch.cat([skip, up], dim=1) (task_decoders.py:49) without materialising the cat.
//
// T = bf16: operands are staged as they are (throughput mode).
// T = float ("split" mode): every fp32 operand is staged as three bf16 images h = bf16(x),
//   m = bf16(x - h), l = bf16(x - h - m) (24 mantissa bits in total) and each product is formed on the
//   same MFMA path from the six partial products of order <= 2^-18 (hh, hm, mh, hl, lh, mm), smallest
//   first, fp32 accumulation: fp32-grade products at 6/16 of the cost of the f32 MFMA.  This is the
//   parity mode that is checked against the fp32 oracle.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

// Timing ablations that change results (skip a DMA, the MFMAs, the stores ...) and the superseded forward loops are
// compiled only with -DS2S_ABLATE (stain2stain_amd/_native.py build(ablate=True) -> libstain2stain_hip_ablate.so, used by
// scripts/ alone); the product library cannot be switched into them by an environment variable.
#ifdef S2S_ABLATE
#define S2S_ABL(cond) (cond)
#define S2S_DBG_MASK (~0)
#else
#define S2S_ABL(cond) false
#define S2S_DBG_MASK (64 | 128)
#endif

struct Conv3x3Args {
  const void* x0;
  const void* x1;
  const void* w;        // packed [nchunk][9][Cout][32]
  const float* bias;    // [Cout] or null
  void* y;
  float* stat_part;     // [2][Cout][gridDim.x] (channel-major: the finalize kernel reads each channel's row contiguously) or null
  const float* ep_scale;  // optional epilogue affine (eval-mode BatchNorm folded), [Cout]
  const float* ep_shift;
  int ld0, c0, ld1, c1, ldy;
  int B, H, W, Cout, tilesY, tilesX, nchunk;
  int relu;
  // activation of a layer WITHOUT a norm behind it (row a13: the pix2pix generator's outermost / innermost layers, the
  // PatchGAN's first layer), applied to bias-added outputs when no folded affine is given: act = 1 -> t > 0 ? t :
  // act_slope * t (LeakyReLU; slope 0 = ReLU).  y2 (optional): a second copy max(y, 0) with its own pixel stride --
  // the ReLU'd skip tensor written straight into the decoder's concatenation buffer.
  int act;
  float act_slope;
  void* y2;
  int ldy2;
  // split-K (convkxk_dma16_kernel only): gridDim.z workgroups share one output tile, each reduces a contiguous range
  // of the 32-channel chunks and writes its fp32 partial tile to kpart[z][B*H*W][Cout]; convk_splitk_reduce_kernel
  // adds them up and applies bias / activation.  nullptr = one workgroup per tile, epilogue in the kernel.
  float* kpart;
  int ksplit;
  int lgc;        // convkxk MODE 1: log2 of the real input channel count (a.c0 = 4 << lgc virtual channels)
  int lgh, lgw;   // convflat_dma16_kernel: log2 of the map size the flattened pixel index is decoded with
  int bias_mod;   // bias index = channel % bias_mod (= Cout normally; Cout / 4 for the transposed 4x4 layers, whose four
                  // sub-pixel channel groups share one bias vector)
  // XCD-aware workgroup order (conv3x3_dma16_kernel): workgroups are dealt to the 8 XCDs round-robin by linear id and
  // every XCD has its own L2, so the grid is cut into xsp x xsn = 8 blocks (pixel-tile ranges x output-channel-tile
  // ranges), one per XCD, walked output-channel tile fastest; xsp = 0 = the plain order.
  int xsp, xsn;
  long stat_rows;  // rows per channel of stat_part (tiles, or tiles x wave rows with the register epilogue)
  int direct_ep;   // 1: launches without statistics store straight from the accumulators (conv_epilogue16_direct); S2S_CONV_EPI=lds: 0
  int stat_carry;  // persistent kernel with one channel tile: a wave row's statistics are carried over its tiles and
                   // written once, row = workgroup * WM + wave row (stat_rows = workgroups * WM)
  long stat_rows_req;   // rows the caller sized stat_part for (0: tiles * WM / tiles, the per-tile forms)
  int dbg;   // S2S_CONV_DBG: 64 = clock probe, 128 = old LDS slot key; -DS2S_ABLATE builds only: bit0 = no weight DMA in the
             // 32x32x16 loop, bit1 = no MFMA, bit2 = no halo DMA, 8 = no global stores, 16 = no epilogue
};

// launches without statistics store straight from the accumulators (conv_epilogue16_direct; the LDS-staged epilogue stays
// for the launches that take BatchNorm partials through it)
static int direct_ep_default() { return 1; }

namespace {

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// compile-time loop: the body receives the index as an integral_constant (usable as a template argument)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int ROWB = 80;  // LDS bytes per 32-channel row (64 data + 16 pad)
constexpr int STEM_HALO_BYTES = 3 * 18 * 18 * 4;   // fp32 halo of a 16x16 tile, up to 3 input channels

template <typename T> struct Piece;
template <> struct Piece<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.0f;
  }
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void to_lds(char* hi, int /*img_stride*/, int off) const {
    *reinterpret_cast<bf16x8*>(hi + off) = v;
  }
};
template <> struct Piece<float> {
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

// ---- epilogue shared by both main loops ---------------------------------------------------------------
// bias / folded affine / ReLU, per-channel partial statistics, coalesced store.
template <typename T, int TH, int TW, int BN, int WM, int WN, int LDS_MAIN>
__device__ __forceinline__ void conv_epilogue(const Conv3x3Args& a, f32x16 (&acc)[(TH * TW / WM) / 32][(BN / WN) / 32],
                                              char* smem, int img, int y0, int x0p, int n0) {
  constexpr bool SPLIT = std::is_same<T, float>::value;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
  T* __restrict__ yout = static_cast<T*>(a.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  // The accumulator tile goes through LDS (free after the last barrier) so that the global stores are
  // whole pixel rows: 16 B per lane, BN*sizeof(T) contiguous bytes per pixel, instead of 2-byte scatters.
  constexpr int RS = BN * (int)sizeof(T) + (SPLIT ? 16 : 64);   // out-tile row stride (bytes), bank-staggered
  // the staging tile must not be larger than the main-loop LDS (it would cost a resident workgroup): big
  // tiles are flushed in EP passes, one group of WM/EP wave rows at a time
  constexpr int EP = (BM * RS + WM * 2 * BN * 4 <= LDS_MAIN) ? 1 : ((BM / 2) * RS + WM * 2 * BN * 4 <= LDS_MAIN ? 2 : 4);
  static_assert(WM % EP == 0 || EP == 1, "epilogue passes split the wave rows");
  constexpr int PM = BM / EP;                                         // pixels per pass
  char* const otile = smem;                                           // [PM][RS]
  float* const red = reinterpret_cast<float*>(smem + PM * RS);       // [WM][2][BN]
  const bool want_stats = a.stat_part != nullptr;
  float s1[NI], s2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) { s1[ni] = 0.f; s2[ni] = 0.f; }
#pragma unroll
  for (int ep = 0; ep < EP; ++ep) {
    if (ep > 0) __syncthreads();
    if (wm / (WM / EP) == ep || EP == 1) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int nl = wn * WTN + ni * 32 + r;
        const int n = n0 + nl;
        const bool nok = n < a.Cout;
        const float bias = (nok && a.bias) ? a.bias[n % a.bias_mod] : 0.f;
        const float esc = (nok && a.ep_scale) ? a.ep_scale[n] : 1.f;
        const float esh = (nok && a.ep_shift) ? a.ep_shift[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
            const int m = wm * WTM + mi * 32 + row;
            const int py = m / TW, px = m - py * TW;
            float t = acc[mi][ni][j] + bias;
            if (a.ep_scale) t = t * esc + esh;
            if (a.relu) t = fmaxf(t, 0.f);
            if (a.act) t = t > 0.f ? t : a.act_slope * t;
            v[j] = t;
            if (nok && y0 + py < a.H && x0p + px < a.W) { s1[ni] += t; s2[ni] += t * t; }
          }
          const int mrow0 = wm * WTM + mi * 32 + 4 * h - ep * PM;
          if constexpr (SPLIT) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
              *reinterpret_cast<float*>(otile + (mrow0 + (j & 3) + 8 * (j >> 2)) * RS + nl * 4) = v[j];
          } else {
            // lanes (2k, 2k+1) hold channels (n, n+1) of the same rows: swap so each lane owns a channel
            // PAIR of one row and writes one dword -- even lane row j, odd lane row j+1
            const bool odd = lane & 1;
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
              const float got = __shfl_xor(odd ? v[j] : v[j + 1], 1, 64);
              bf16x2 pk;
              pk[0] = (bf16_t)(odd ? got : v[j]);
              pk[1] = (bf16_t)(odd ? v[j + 1] : got);
              const int mr = mrow0 + ((j + (odd ? 1 : 0)) & 3) + 8 * (j >> 2);
              *reinterpret_cast<bf16x2*>(otile + mr * RS + (nl & ~1) * 2) = pk;
            }
          }
        }
      }
    }
    __syncthreads();
    constexpr int EPC = 16 / (int)sizeof(T);          // elements per 16-B chunk
    constexpr int CPR = BN / EPC;                     // chunks per pixel row
    constexpr int O_IT = (PM * CPR + 255) / 256;
#pragma unroll
    for (int i = 0; i < O_IT; ++i) {
      const int idx = tid + i * 256;
      const int ml = idx / CPR, c = idx - ml * CPR;
      const int m = ml + ep * PM;
      const int py = m / TW, px = m - py * TW;
      const int gy = y0 + py, gx = x0p + px, n = n0 + c * EPC;
      if (idx < PM * CPR && gy < a.H && gx < a.W && n < a.Cout) {
        const f32x4 val = *reinterpret_cast<const f32x4*>(otile + ml * RS + c * 16);
        const long opix = ((long)img * a.H + gy) * a.W + gx;
        *reinterpret_cast<f32x4*>(yout + opix * a.ldy + n) = val;
        if (a.y2) {
          if constexpr (SPLIT) {
            f32x4 r4;
#pragma unroll
            for (int k = 0; k < 4; ++k) r4[k] = fmaxf(val[k], 0.f);
            *reinterpret_cast<f32x4*>(static_cast<T*>(a.y2) + opix * a.ldy2 + n) = r4;
          } else {
            const bf16x8 v8 = __builtin_bit_cast(bf16x8, val);
            bf16x8 r8;
#pragma unroll
            for (int k = 0; k < 8; ++k) r8[k] = (float)v8[k] > 0.f ? v8[k] : (bf16_t)0.f;
            *reinterpret_cast<bf16x8*>(static_cast<T*>(a.y2) + opix * a.ldy2 + n) = r8;
          }
        }
      }
    }
  }
  if (want_stats) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nl = wn * WTN + ni * 32 + r;
      const float t1 = s1[ni] + __shfl_xor(s1[ni], 32, 64);
      const float t2 = s2[ni] + __shfl_xor(s2[ni], 32, 64);
      if (h == 0) {
        red[(wm * 2 + 0) * BN + nl] = t1;
        red[(wm * 2 + 1) * BN + nl] = t2;
      }
    }
    __syncthreads();
  }
  if (want_stats) {
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, nl = i - which * BN;
      if (n0 + nl < a.Cout) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WM; ++k) s += red[(k * 2 + which) * BN + nl];
        a.stat_part[((long)which * a.Cout + n0 + nl) * gridDim.x + blockIdx.x] = s;
      }
    }
  }
}

// KS / PAD generalise the loop to the KS x KS-tap forms of row a13 (see convkxk_dma16_kernel below for the geometry:
// a.H / a.W are the OUTPUT size, the input is (H + KS - 1 - 2 PAD) high): this register-staged form is what the fp32
// parity mode of those layers runs on.
template <typename T, int TH, int TW, int BN, int WM, int WN, int KS = 3, int PAD = 1>
__global__ __launch_bounds__(256, (std::is_same<T, float>::value ? 1 : 2)) void conv3x3_mfma_kernel(Conv3x3Args a) {
  constexpr bool SPLIT = std::is_same<T, float>::value;
  constexpr int NIMG = SPLIT ? 3 : 1;
  constexpr int TAPS = KS * KS;
  constexpr int NSET = (TAPS % 3 == 0) ? 3 : 4;      // rotating weight register sets; TAPS % NSET == 0 keeps set = tap % NSET
  static_assert(TAPS % NSET == 0, "weight register rotation");
  constexpr int HW_ = TW + KS - 1, HH_ = TH + KS - 1, HALO = HW_ * HH_;
  constexpr int A_BYTES = HALO * ROWB, B_BYTES = BN * ROWB;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
  constexpr int A_PIECES = HALO * 4, A_IT = (A_PIECES + 255) / 256;
  constexpr int B_PIECES = BN * 4, B_IT = (B_PIECES + 255) / 256;
  static_assert(WM * WN == 4, "four waves per workgroup");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem;                               // [2][NIMG][A_BYTES]
  char* const ldsB = smem + 2 * NIMG * A_BYTES;          // [2][NIMG][B_BYTES]

  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);
  const T* __restrict__ wp = static_cast<const T*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  int bt = blockIdx.x;
  const int tx = bt % a.tilesX; bt /= a.tilesX;
  const int ty = bt % a.tilesY;
  const int img = bt / a.tilesY;
  const int y0 = ty * TH, x0p = tx * TW;
  const int n0 = blockIdx.y * BN;
  const int ctot = a.c0 + a.c1;
  const int Hi = a.H + KS - 1 - 2 * PAD, Wi = a.W + KS - 1 - 2 * PAD;      // input size (= output size for 3x3 / pad 1)

  // ---- per-thread staging coordinates (chunk independent) ----
  int apix[A_IT];    // pixel index into the NHWC tensor, -1 = zero fill
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int idx = tid + i * 256;
    const int px = idx >> 2, pc = idx & 3;
    const int hy = px / HW_, hx = px - hy * HW_;
    const int gy = y0 - PAD + hy, gx = x0p - PAD + hx;
    apix[i] = (idx < A_PIECES && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi)
                  ? (img * Hi + gy) * Wi + gx : -1;
  }
  Piece<T> hreg[A_IT];
  Piece<T> wreg[NSET][B_IT];   // rotating sets (slab index mod NSET == tap mod NSET): NSET - 1 taps of load latency

  auto load_halo = [&](int c) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int ch = c * 32 + ((tid + i * 256) & 3) * 8;
      hreg[i].zero();
      if (apix[i] >= 0) {
        if (ch < a.c0) hreg[i].load(x0 + (long)apix[i] * a.ld0 + ch);
        else if (ch < ctot) hreg[i].load(x1 + (long)apix[i] * a.ld1 + (ch - a.c0));
      }
    }
  };
  auto store_halo = [&](int buf) {
    char* hi = ldsA + buf * NIMG * A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = tid + i * 256;
      if (idx < A_PIECES) hreg[i].to_lds(hi, A_BYTES, (idx >> 2) * ROWB + (idx & 3) * 16);
    }
  };
  auto load_w = [&](int it, auto setc) {
    constexpr int set = decltype(setc)::value;
    const T* base = wp + ((long)it * a.Cout) * 32;   // slabs are packed in iteration order [chunk][tap]
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 2, pc = idx & 3;
      wreg[set][i].zero();
      if (idx < B_PIECES && n0 + row < a.Cout) wreg[set][i].load(base + (long)(n0 + row) * 32 + pc * 8);
    }
  };
  auto store_w = [&](int buf, auto setc) {
    constexpr int set = decltype(setc)::value;
    char* hi = ldsB + buf * NIMG * B_BYTES;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int idx = tid + i * 256;
      if (idx < B_PIECES) wreg[set][i].to_lds(hi, B_BYTES, (idx >> 2) * ROWB + (idx & 3) * 16);
    }
  };

  // ---- fragment base offsets ----
  int abase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 32 + r;
    const int py = m / TW, px = m - py * TW;
    abase[mi] = (py * HW_ + px) * ROWB + h * 16;
  }
  int bbase[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bbase[ni] = (wn * WTN + ni * 32 + r) * ROWB + h * 16;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

  const int nit = a.nchunk * TAPS;

  // ---- prologue ----
  load_w(0, std::integral_constant<int, 0>{});
  load_halo(0);
  store_w(0, std::integral_constant<int, 0>{});
  store_halo(0);
  static_for<NSET - 1>([&](auto k) {
    constexpr int kk = decltype(k)::value + 1;
    if (nit > kk) load_w(kk, std::integral_constant<int, kk>{});
  });
  __syncthreads();

  for (int c = 0; c < a.nchunk; ++c) {
    if (c + 1 < a.nchunk) load_halo(c + 1);
    const char* Ahi = ldsA + (c & 1) * NIMG * A_BYTES;
    static_for<TAPS>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      const int it = c * TAPS + tap;
      const char* Bhi = ldsB + (it & 1) * NIMG * B_BYTES;
      const int tapoff = ((tap / KS) * HW_ + (tap % KS)) * ROWB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[MI], bfr[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          af[mi] = *reinterpret_cast<const bf16x8*>(Ahi + abase[mi] + tapoff + ks * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bfr[ni] = *reinterpret_cast<const bf16x8*>(Bhi + bbase[ni] + ks * 32);
        if constexpr (SPLIT) {
          bf16x8 am[MI], al[MI], bm[NI], bl[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            am[mi] = *reinterpret_cast<const bf16x8*>(Ahi + A_BYTES + abase[mi] + tapoff + ks * 32);
            al[mi] = *reinterpret_cast<const bf16x8*>(Ahi + 2 * A_BYTES + abase[mi] + tapoff + ks * 32);
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            bm[ni] = *reinterpret_cast<const bf16x8*>(Bhi + B_BYTES + bbase[ni] + ks * 32);
            bl[ni] = *reinterpret_cast<const bf16x8*>(Bhi + 2 * B_BYTES + bbase[ni] + ks * 32);
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bm[ni], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bl[ni], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bm[ni], acc[mi][ni], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
      }
      // slot (it+1)&1 was last read by tap it-1, which every wave left at the previous barrier
      // slab it+1 sits in register set (tap+1)%NSET (loaded NSET-1 taps ago); slab it+NSET goes into the set that
      // held slab it (tap%NSET), whose LDS copy was written one tap ago
      if (it + 1 < nit) store_w((it + 1) & 1, std::integral_constant<int, (tap + 1) % NSET>{});
      if (it + NSET < nit) load_w(it + NSET, std::integral_constant<int, tap % NSET>{});
      if (tap == TAPS - 1 && c + 1 < a.nchunk) store_halo((c + 1) & 1);
      __syncthreads();
    });
  }

  conv_epilogue<T, TH, TW, BN, WM, WN, 2 * NIMG * (A_BYTES + B_BYTES)>(a, acc, smem, img, y0, x0p, n0);
}

// =========================================================================================================
// bf16 main loop v2: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), no VGPR staging.
//   * LDS rows are 64 B (32 channels), unpadded because a DMA wave-instruction writes 1 KiB linearly
//     (16 rows x 4 pieces); bank conflicts are removed by an XOR swizzle instead: piece p of row r is stored
//     in slot p ^ ((r >> 2) & 3).  For the DMA this swizzle sits on the per-lane SOURCE address (lane l
//     fetches piece (l & 3) ^ ((l >> 4) & 3)); the fragment reads apply the same XOR.  Any 16 consecutive
//     rows then cover the 16 slots of the 256-B bank row exactly once.
//   * zero padding / ragged edges: out-of-range lanes fetch from a 64-B zero page instead of being masked, so
//     every wave issues the same number of DMA instructions and the counted s_waitcnt vmcnt(N) is exact.
//   * weight slabs stream through an NS-slot ring, NS-1 taps ahead of their use; the halo of the next
//     32-channel chunk is fetched during the current chunk.  One raw s_barrier per tap; DMAs stay in flight
//     across it (a __syncthreads() would drain them).
// =========================================================================================================
__device__ __attribute__((aligned(64))) char g_zero_page[64];
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void dma16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef S2S_ABLATE   // the 32x32x16 form of the LDS-DMA loop: ablation builds only (scripts/), never in the product library
template <int TH, int TW, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256, 2) void conv3x3_dma_kernel(Conv3x3Args a) {
  using T = bf16_t;
  // halo image: row index = hy * HP + hx with the pitch HP = TW + 4 a multiple of 4, so that the bank-row phase
  // (row & 3) equals hx & 3 and the swizzle can be keyed on hx alone: slot = piece ^ ((hx >> 2) & 3).  A tap's
  // kh then is a constant byte offset (folded into the ds_read immediate) and only kw changes the swizzle.
  constexpr int HP = TW + 4, HH_ = TH + 2, ROWS = HH_ * HP;
  constexpr int NGA = (ROWS + 15) / 16, HG = (NGA + 3) / 4;   // halo row groups, DMA instr per wave
  constexpr int A_BYTES = HG * 4 * 1024;
  constexpr int BG = BN / 64;                                 // weight DMA instr per wave per slab
  constexpr int B_BYTES = BN * 64;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
  static_assert(WM * WN == 4 && BN % 64 == 0 && (NS == 3 || NS == 4), "configuration");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem;                    // [2][A_BYTES]
  char* const ldsB = smem + 2 * A_BYTES;      // [NS][B_BYTES]

  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);
  const char* __restrict__ wp = static_cast<const char*>(a.w);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  int bt = blockIdx.x;
  const int tx = bt % a.tilesX; bt /= a.tilesX;
  const int ty = bt % a.tilesY;
  const int img = bt / a.tilesY;
  const int y0 = ty * TH, x0p = tx * TW;
  const int n0 = blockIdx.y * BN;
  const int ctot = a.c0 + a.c1;

  // ---- DMA lane geometry ----
  const int drow = lane >> 2, dslot = lane & 3;
  int apix[HG];    // NHWC pixel index of this lane's halo row in DMA group j, -1 = zero page
  int apc[HG];     // channel offset (elements) inside a 32-channel chunk of the piece this lane fetches
#pragma unroll
  for (int j = 0; j < HG; ++j) {
    const int row = (wave + 4 * j) * 16 + drow;
    const int hy = row / HP, hx = row - hy * HP;
    const int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
    apix[j] = (row < ROWS && hx < TW + 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                  ? (img * a.H + gy) * a.W + gx : -1;
    apc[j] = (dslot ^ ((hx >> 2) & 3)) * 8;
  }
  // weights: slab `it` = [Cout][32] at byte offset it * Cout * 64; every lane keeps a running pointer to its
  // row piece (rows past Cout park on the zero page with a zero stride)
  const char* wptr[BG];
  int wstep[BG];
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int n = (wave + 4 * j) * 16 + drow;
    const bool ok = n0 + n < a.Cout;
    wptr[j] = ok ? wp + ((long)(n0 + n) * 32 + ((dslot ^ ((n >> 2) & 3)) * 8)) * 2 : g_zero_page;
    wstep[j] = ok ? a.Cout * 64 : 0;
  }

  auto dma_halo = [&](int c) {
    char* dst = ldsA + (c & 1) * A_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      const int ch = c * 32 + apc[j];
      const void* g = g_zero_page;
      if (apix[j] >= 0) {
        if (ch < a.c0) g = x0 + (long)apix[j] * a.ld0 + ch;
        else if (ch < ctot) g = x1 + (long)apix[j] * a.ld1 + (ch - a.c0);
      }
      dma16(g, dst + j * 4096);
    }
  };
  auto dma_w = [&](int slot) {   // next slab in sequence
    char* dst = ldsB + slot * B_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      dma16(wptr[j], dst + j * 4096);
      wptr[j] += wstep[j];
    }
  };

  // ---- fragment geometry (computed once) ----
  int aofs[MI][3];   // byte offset of this lane's pixel row for tap (0, kw), k-step 0
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 32 + r;
    const int py = m / TW, px = m - py * TW;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
      aofs[mi][kw] = (py * HP + px + kw) * 64 + ((h ^ (((px + kw) >> 2) & 3)) << 4);
  }
  int bofs[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = wn * WTN + ni * 32 + r;
    bofs[ni] = n * 64 + ((h ^ ((n >> 2) & 3)) << 4);
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

  // one tap: MFMAs of tap `tap` from halo buffer Ab and weight slot (it % NS)
  auto compute = [&](auto tapc, const char* Ab, const char* Bb) {
    constexpr int tap = decltype(tapc)::value;
    constexpr int kh = tap / 3, kw = tap % 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const bf16x8*>(Ab + (aofs[mi][kw] ^ (ks << 5)) + kh * HP * 64);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(Bb + (bofs[ni] ^ (ks << 5)));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    }
  };

  // ---- prologue: halo 0 and the first NS-1 slabs (slabs past the end are harmless: they read the next
  //      chunk's bytes of the packed weights only if they exist, so clamp by count) ----
  const int nit = a.nchunk * 9;
  dma_halo(0);
  static_for<NS - 1>([&](auto k) { dma_w(decltype(k)::value); });   // nit >= 9 > NS-1
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  // ---- steady state: every chunk but the last; no run-time conditions inside a tap ----
  int c = 0;
  for (; c + 1 < a.nchunk; ++c) {
    const char* Ab = ldsA + (c & 1) * A_BYTES;
    const int it0 = c * 9;
    static_for<9>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap == 0) { if (!(a.dbg & 4)) dma_halo(c + 1); else { static_for<HG>([&](auto) { dma16(g_zero_page, ldsA + ((c + 1) & 1) * A_BYTES + wave * 1024); }); } }
      if (!(a.dbg & 1)) dma_w((it0 + tap + NS - 1) % NS);            // the slot read one tap ago
      else { static_for<BG>([&](auto) { dma16(g_zero_page, ldsB + wave * 1024); }); }
      if (!(a.dbg & 2)) compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);           // this tap's MFMAs (and LDS reads) stay ahead of the barrier
      // retire slab it+1: the (NS-2) younger slabs, and the halo while it is younger, may stay in flight
      wait_vm<(NS - 2) * BG + (tap <= NS - 3 ? HG : 0)>();
      __builtin_amdgcn_s_barrier();
    });
  }
  // ---- last chunk: no further halo; slabs run out NS-1 taps before the end ----
  {
    const char* Ab = ldsA + (c & 1) * A_BYTES;
    const int it0 = c * 9;
    static_for<9>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap + NS - 1 < 9) dma_w((it0 + tap + NS - 1) % NS);
      compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + NS - 1 < 9) wait_vm<(NS - 2) * BG>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
    });
  }
  conv_epilogue<T, TH, TW, BN, WM, WN, 2 * A_BYTES + NS * B_BYTES>(a, acc, smem, img, y0, x0p, n0);
}

#endif  // S2S_ABLATE

// =========================================================================================================
// bf16 main loop v3: the LDS-DMA loop above on v_mfma_f32_16x16x32_bf16.  One MFMA covers the whole 32-channel
// chunk of a tap (K = 32), a lane's A/B fragment is one 16-B piece (row = lane & 15, piece = lane >> 4), and the
// result tile has col = lane & 15, rows 4*(lane >> 4) + reg.  Same FLOP per cycle as the 32x32x16 form, but the
// chip holds a higher clock on it under load (MI355X_MICROARCH.md, DVFS give-back 7), which is what bounds the
// conv stack on real data.  The swizzle uses slot = piece ^ ((-(row >> 2)) & 3): with the 16x16 lane->row map a
// ds_read_b128 lane group takes rows {0-3, 12-15} of one piece and rows {4-11} of the next, and this key keeps
// all sixteen 16-B slots of the bank row distinct (the plain (row >> 2) & 3 key gives a 2-way conflict here).
// =========================================================================================================
// Epilogue of the 16x16x32 kernel.  The main loop issues its MFMAs with the operands swapped (weights as A, pixels
// as B), so a lane's four accumulator registers are four consecutive output CHANNELS of one pixel: they pack into
// one 8-byte LDS write of the pixel-major tile (the un-swapped layout needed a lane-pair shuffle and two 4-byte
// writes per fragment, and that epilogue cost 5.4 us per workgroup -- a third of a 64->64 layer).  The tile is then
// read back 16 B per thread for coalesced NHWC stores; BatchNorm's partial sums are taken there, from the values as
// stored (bf16), which is what a BatchNorm behind a bf16 conv normalises.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for every global
// store a wave has issued to be acknowledged (1-2 us in the epilogue, where the stores are fire-and-forget).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// Where a tile's output pixel (gy, gx) goes: plain (os = 1) or, for the sub-pixel phases of a transposed convolution,
// pixel (os gy + oy, os gx + ox) of the OH x OW output.
struct OutMap { int os, oy, ox, OH, OW; };

// `tid` is the thread's index inside its group of 256 (a whole workgroup, or one half of a ping-pong workgroup),
// `smem` that group's staging area, `stat_row` the tile's row in stat_part; the barriers are workgroup-wide.
template <int TH, int TW, int BN, int WM, int WN, int LDS_MAIN, bool AFFINE>
__device__ __forceinline__ void conv_epilogue16(const Conv3x3Args& a, f32x4 (&acc)[(TH * TW / WM) / 16][(BN / WN) / 16],
                                                const float (&bias)[(BN / WN) / 16][4], char* smem, int img, int y0,
                                                int x0p, int n0, int tid, long stat_row, const OutMap om) {
  using T = bf16_t;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  T* __restrict__ yout = static_cast<T*>(a.y);
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int cl = lane & 15, q = lane >> 4;
  constexpr int RS = BN * 2 + 16;                          // row stride 4 (mod 64) dwords: 8-byte writes 2-way at worst
  constexpr int CPR = BN / 8, RG = 256 / CPR;              // 16-byte pieces per pixel row, row groups of the read-back
  constexpr int RED = 4 * 2 * BN * 4;                      // per-wave partial sums
  constexpr int EP = (BM * RS + RED <= LDS_MAIN) ? 1 : ((BM / 2) * RS + RED <= LDS_MAIN ? 2 : 4);
  static_assert(WM % EP == 0 || EP == 1, "epilogue passes split the wave rows");
  static_assert((BM / EP) * RS + RED <= LDS_MAIN, "epilogue staging fits the main-loop LDS");
  constexpr int PM = BM / EP;
  char* const otile = smem;
  float* const red = reinterpret_cast<float*>(smem + PM * RS);
  const bool want_stats = a.stat_part != nullptr;
  const bool full = y0 + TH <= a.H && x0p + TW <= a.W && n0 + BN <= a.Cout;
  // channel constants of this lane: channels wn*WTN + ni*16 + 4q + j (the bias was fetched before the last chunk)
  float esc[NI][4], esh[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * WTN + ni * 16 + 4 * q + j;
      const bool nok = n < a.Cout;
      esc[ni][j] = (AFFINE && nok) ? a.ep_scale[n] : 1.f;
      esh[ni][j] = (AFFINE && nok) ? a.ep_shift[n] : 0.f;
    }
  const int c = tid % CPR, rg = tid / CPR;                 // read-back: fixed 8-channel piece, rows rg, rg+RG, ...
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
  for (int ep = 0; ep < EP; ++ep) {
    if (ep > 0) lds_barrier();
    if (wm / (WM / EP) == ep || EP == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int mrow = wm * WTM + mi * 16 + cl - ep * PM;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bf16x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = acc[mi][ni][j] + bias[ni][j];
            if (AFFINE) {
              t = t * esc[ni][j] + esh[ni][j];
              if (a.relu) t = fmaxf(t, 0.f);
            } else if (a.act) {
              t = t > 0.f ? t : a.act_slope * t;
            }
            pk[j] = (bf16_t)t;
          }
          *reinterpret_cast<bf16x4*>(otile + mrow * RS + (wn * WTN + ni * 16 + 4 * q) * 2) = pk;
        }
      }
    }
    lds_barrier();
    constexpr int O_IT = PM / RG;
    static_assert(PM % RG == 0, "read-back rows split evenly");
#pragma unroll
    for (int i = 0; i < O_IT; ++i) {
      const int ml = rg + i * RG;
      const int m = ml + ep * PM;
      const int py = m / TW, px = m - py * TW;
      const int gy = y0 + py, gx = x0p + px, n = n0 + c * 8;
      if (full || (gy < a.H && gx < a.W && n < a.Cout)) {
        const bf16x8 val = *reinterpret_cast<const bf16x8*>(otile + ml * RS + c * 16);
        const long opix = ((long)img * om.OH + gy * om.os + om.oy) * om.OW + gx * om.os + om.ox;
        if (!S2S_ABL(a.dbg & 8)) *reinterpret_cast<bf16x8*>(yout + opix * a.ldy + n) = val;
        if (a.y2) {
          bf16x8 r8;
#pragma unroll
          for (int k = 0; k < 8; ++k) r8[k] = (float)val[k] > 0.f ? val[k] : (bf16_t)0.f;
          *reinterpret_cast<bf16x8*>(static_cast<T*>(a.y2) + opix * a.ldy2 + n) = r8;
        }
        if (want_stats) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float t = (float)val[k];
            s1[k] += t;
            s2[k] = fmaf(t, t, s2[k]);
          }
        }
      }
    }
  }
  if (want_stats) {
    // the 64 / CPR row groups of a wave are folded with shuffles, then red[wave][which][BN] across the four waves;
    // channels past Cout hold zeros (their rows were skipped above) and are not written out
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int m = CPR; m < 64; m <<= 1) { s1[k] += __shfl_xor(s1[k], m, 64); s2[k] += __shfl_xor(s2[k], m, 64); }
    if (lane < CPR) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(wave * 2 + 0) * BN + c * 8 + k] = s1[k];
        red[(wave * 2 + 1) * BN + c * 8 + k] = s2[k];
      }
    }
    lds_barrier();
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, nl = i - which * BN;
      if (n0 + nl < a.Cout && stat_row >= 0)
        a.stat_part[((long)which * a.Cout + n0 + nl) * a.stat_rows + stat_row] =
            (red[(0 * 2 + which) * BN + nl] + red[(1 * 2 + which) * BN + nl]) +
            (red[(2 * 2 + which) * BN + nl] + red[(3 * 2 + which) * BN + nl]);
    }
  }
}

// Epilogue of the 16x16x32 kernels for launches WITHOUT statistics (every data gradient, the eval-mode forward, the 4x4
// layers of row a13): straight from the accumulators to global memory, no LDS staging, no barrier.  A lane holds four
// consecutive channels of one pixel per 16 x 16 block; v_permlane16_swap exchanges the odd 16-lane rows of block ni with
// the even rows of block ni + 1, after which every lane owns EIGHT consecutive channels of its pixel -- one 16-byte
// store, and the four lanes of a pixel cover 64 contiguous bytes (two stores complete a 128-byte line of a 64-channel
// tile).  (The LDS-staged form spends ~3 us per workgroup on its write / barrier / read-back; a 64 -> 64 layer at 256^2
// has 18 taps = 6.8 us of main loop per tile, and the data gradients take no statistics: measured 40 of 132 us there.)
// STATS: BatchNorm's per-channel (sum, sum of squares) of the values AS STORED (bf16), taken in registers: after the
// exchange a lane's eight channels are the same for all of its pixels, so it sums over its MI pixels, the sixteen pixel
// lanes of a row are folded with four DPP row rotations, and lane 0 of each row writes the wave row's partial sums to
// stat_part[which][channel][stat_row] -- one row per (tile, wave row), no LDS, no barrier (s2s_bn_finalize sums the rows
// of a channel whatever their number).
template <int BITS>
__device__ __forceinline__ float row_ror_add(float v) {
  // v + (v rotated right by BITS lanes inside its 16-lane row): DPP row_ror
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + BITS, 0xf, 0xf, false));
}

// Fold a wave's per-lane statistics over the sixteen pixel lanes of each row (four DPP row rotations) and write the wave
// row's partial sums: stat_part[which][channel][stat_row].
template <int BN, int WN>
__device__ __forceinline__ void conv_stats_flush(const Conv3x3Args& a, float (*s1)[8], float (*s2)[8], int n0, int tid,
                                                 long stat_row) {
  constexpr int WTN = BN / WN, NI = WTN / 16;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, cl = lane & 15, q = lane >> 4;
  const int nlane = n0 + wn * WTN + (q & 1) * 16 + (q >> 1) * 8;
#pragma unroll
  for (int pr = 0; pr < NI / 2; ++pr)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float u = s1[pr][k], w2 = s2[pr][k];
      u = row_ror_add<8>(u); u = row_ror_add<4>(u); u = row_ror_add<2>(u); u = row_ror_add<1>(u);
      w2 = row_ror_add<8>(w2); w2 = row_ror_add<4>(w2); w2 = row_ror_add<2>(w2); w2 = row_ror_add<1>(w2);
      s1[pr][k] = u; s2[pr][k] = w2;
    }
  if (cl == 0 && stat_row >= 0) {
#pragma unroll
    for (int pr = 0; pr < NI / 2; ++pr)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int n = nlane + pr * 32 + k;
        if (n < a.Cout) {
          a.stat_part[((long)n) * a.stat_rows + stat_row] = s1[pr][k];
          a.stat_part[((long)a.Cout + n) * a.stat_rows + stat_row] = s2[pr][k];
        }
      }
  }
}

// CARRY (with STATS): the per-lane sums live in the CALLER's registers (cs1 / cs2, zeroed by it) and are neither folded
// nor stored here -- a persistent workgroup carries them over all of its tiles and calls conv_stats_flush once.
template <int TH, int TW, int BN, int WM, int WN, bool AFFINE, bool STATS = false, bool CARRY = false>
__device__ __forceinline__ void conv_epilogue16_direct(const Conv3x3Args& a, f32x4 (&acc)[(TH * TW / WM) / 16][(BN / WN) / 16],
                                                       const float (&bias)[(BN / WN) / 16][4], int img, int y0, int x0p,
                                                       int n0, int tid, const OutMap om, long stat_row = -1,
                                                       float (*cs1)[8] = nullptr, float (*cs2)[8] = nullptr) {
  using T = bf16_t;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  static_assert(NI % 2 == 0, "channel blocks are exchanged in pairs");
  T* __restrict__ yout = static_cast<T*>(a.y);
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int cl = lane & 15, q = lane >> 4;
  float esc[NI][4], esh[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * WTN + ni * 16 + 4 * q + j;
      const bool nok = n < a.Cout;
      esc[ni][j] = (AFFINE && nok) ? a.ep_scale[n] : 1.f;
      esh[ni][j] = (AFFINE && nok) ? a.ep_shift[n] : 0.f;
    }
  // after the exchange this lane holds channels [blk * 16 + (q >> 1) * 8, + 8) with blk = ni + (q & 1)
  const int nlane = n0 + wn * WTN + (q & 1) * 16 + (q >> 1) * 8;
  float s1l[STATS ? NI / 2 : 1][8], s2l[STATS ? NI / 2 : 1][8];
  float (*const s1)[8] = CARRY ? cs1 : s1l;
  float (*const s2)[8] = CARRY ? cs2 : s2l;
  if constexpr (STATS && !CARRY) {
#pragma unroll
    for (int pr = 0; pr < NI / 2; ++pr)
#pragma unroll
      for (int k = 0; k < 8; ++k) { s1[pr][k] = 0.f; s2[pr][k] = 0.f; }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 16 + cl;
    const int py = m / TW, px = m - py * TW;
    const int gy = y0 + py, gx = x0p + px;
    const bool pok = gy < a.H && gx < a.W;
    const long opix = ((long)img * om.OH + gy * om.os + om.oy) * om.OW + gx * om.os + om.ox;
#pragma unroll
    for (int ni = 0; ni < NI; ni += 2) {
      unsigned w[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        bf16x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = acc[mi][ni + b][j] + bias[ni + b][j];
          if (AFFINE) {
            t = t * esc[ni + b][j] + esh[ni + b][j];
            if (a.relu) t = fmaxf(t, 0.f);
          } else if (a.act) {
            t = t > 0.f ? t : a.act_slope * t;
          }
          pk[j] = (bf16_t)t;
        }
        const uint2 u = __builtin_bit_cast(uint2, pk);
        w[b][0] = u.x; w[b][1] = u.y;
      }
      const auto r0 = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
      const uint4 out = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      const int n = nlane + ni * 16;
      if constexpr (STATS) {
        if (pok && n < a.Cout) {
          const bf16x8 val = __builtin_bit_cast(bf16x8, out);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float t = (float)val[k];
            s1[ni / 2][k] += t;
            s2[ni / 2][k] = fmaf(t, t, s2[ni / 2][k]);
          }
        }
      }
      if (pok && n < a.Cout) {
        if (!S2S_ABL(a.dbg & 8)) *reinterpret_cast<uint4*>(yout + opix * a.ldy + n) = out;
        if (a.y2) {
          const bf16x8 val = __builtin_bit_cast(bf16x8, out);
          bf16x8 r8;
#pragma unroll
          for (int k = 0; k < 8; ++k) r8[k] = (float)val[k] > 0.f ? val[k] : (bf16_t)0.f;
          *reinterpret_cast<bf16x8*>(static_cast<T*>(a.y2) + opix * a.ldy2 + n) = r8;
        }
      }
    }
  }
  if constexpr (STATS && !CARRY) conv_stats_flush<BN, WN>(a, s1, s2, n0, tid, stat_row);
}

// S2S_CONV_DBG=64: thread 0 of each workgroup records the shader-clock counter and the 100 MHz wall clock at entry
// and exit (read back with s2s_debug_conv_clock): the clock the kernel actually ran at under its own load.
__device__ long g_clk[8192 * 4];
template <int TH, int TW, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256, 2) void conv3x3_dma16_kernel(Conv3x3Args a) {
  long c0_ = 0, w0_ = 0;
  if ((a.dbg & 64) && threadIdx.x == 0) { c0_ = __builtin_readcyclecounter(); w0_ = wall_clock64(); }
  using T = bf16_t;
  constexpr int HP = TW + 4, HH_ = TH + 2, ROWS = HH_ * HP;
  constexpr int NGA = (ROWS + 15) / 16, HG = (NGA + 3) / 4;
  constexpr int A_BYTES = HG * 4 * 1024;
  constexpr int BG = BN / 64;
  constexpr int B_BYTES = BN * 64;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int RB = TW / 16;                        // 16-pixel blocks per tile row
  static_assert(WM * WN == 4 && BN % 64 == 0 && NS >= 3 && NS <= 8 && TW % 16 == 0 && WTM % TW == 0, "configuration");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem;
  char* const ldsB = smem + 2 * A_BYTES;

  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);
  const char* __restrict__ wp = static_cast<const char*>(a.w);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int cl = lane & 15, kp = lane >> 4;           // fragment row / 16-B piece (k group)

  int bt = blockIdx.x, by = blockIdx.y;
  if (a.xsp) {       // this workgroup runs on XCD (linear id % 8): give it a tile of that XCD's block of the grid
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = L & 7, k = L >> 3;
    const unsigned xp = xcd % (unsigned)a.xsp, xn = xcd / (unsigned)a.xsp;
    const unsigned P = gridDim.x / (unsigned)a.xsp, Nn = gridDim.y / (unsigned)a.xsn;
    const unsigned kp_ = k / Nn, kn_ = k - kp_ * Nn;
    bt = (int)(xp * P + kp_); by = (int)(xn * Nn + kn_);
  }
  const int tile_id = bt;
  const int tx = bt % a.tilesX; bt /= a.tilesX;
  const int ty = bt % a.tilesY;
  const int img = bt / a.tilesY;
  const int y0 = ty * TH, x0p = tx * TW;
  const int n0 = by * BN;
  const int ctot = a.c0 + a.c1;
  // split-K (small batches: sampling one tile at a time leaves a 16x16 level with 32 workgroups and 288 taps each):
  // gridDim.z workgroups share the tile, each takes a range of the chunks and leaves an fp32 partial tile
  const int c_lo = (int)(((long)blockIdx.z * a.nchunk) / gridDim.z), c_hi = (int)(((long)(blockIdx.z + 1) * a.nchunk) / gridDim.z);

  const int drow = lane >> 2, dslot = lane & 3;
  // 16-byte slot of a halo pixel's 64-byte row: piece ^ key(halo column).  The three taps of a kernel row read the
  // fragments at columns c, c + 1, c + 2, and a ds_read_b128 lane group mixes rows {0-3, 12-15} of one piece with rows
  // {4-11} of the next: the key 2 * ((column >> 2) & 1) keeps the sixteen slots of every such group distinct for all
  // three shifts (exhaustive check over shifts, row blocks and lane groups); the earlier (-(column >> 2)) & 3 was only
  // conflict-free unshifted, and SQ_LDS_BANK_CONFLICT showed 29 % of this kernel's LDS cycles as conflicts
  // (profiles/r02_f_sq_counters.json; S2S_CONV_DBG=128 selects the old key).
  const bool old_key = (a.dbg & 128) != 0;
  auto akey = [&](int col) { return old_key ? ((-(col >> 2)) & 3) : (((col >> 2) & 1) << 1); };
  int apix[HG], apc[HG];
#pragma unroll
  for (int j = 0; j < HG; ++j) {
    const int row = (wave + 4 * j) * 16 + drow;
    const int hy = row / HP, hx = row - hy * HP;
    const int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
    apix[j] = (row < ROWS && hx < TW + 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                  ? (img * a.H + gy) * a.W + gx : -1;
    apc[j] = (dslot ^ akey(hx)) * 8;
  }
  const char* wptr[BG];
  int wstep[BG];
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int n = (wave + 4 * j) * 16 + drow;
    const bool ok = n0 + n < a.Cout;
    wptr[j] = ok ? wp + (long)c_lo * 9 * a.Cout * 64 + ((long)(n0 + n) * 32 + ((dslot ^ ((-(n >> 2)) & 3)) * 8)) * 2 : g_zero_page;
    wstep[j] = ok ? a.Cout * 64 : 0;
  }
  auto dma_halo = [&](int c) {
    char* dst = ldsA + (c & 1) * A_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      const int ch = c * 32 + apc[j];
      const void* g = g_zero_page;
      if (apix[j] >= 0) {
        if (ch < a.c0) g = x0 + (long)apix[j] * a.ld0 + ch;
        else if (ch < ctot) g = x1 + (long)apix[j] * a.ld1 + (ch - a.c0);
      }
      dma16(g, dst + j * 4096);
    }
  };
  auto dma_w = [&](int slot) {
    char* dst = ldsB + slot * B_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      dma16(wptr[j], dst + j * 4096);
      wptr[j] += wstep[j];
    }
  };

  // fragment offsets: block mi of the wave = tile row (wm*WTM/TW + mi/RB), columns (mi%RB)*16 + lane&15
  int aofs[RB][3];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int px = rb * 16 + cl + kw;
      aofs[rb][kw] = ((wm * (WTM / TW)) * HP + px) * 64 + ((kp ^ akey(px)) << 4);
    }
  const int nrow = wn * WTN + cl;
  const int bofs = nrow * 64 + ((kp ^ ((-(nrow >> 2)) & 3)) << 4);   // + ni*16*64 (same swizzle phase)

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mi][ni][j] = 0.f;

  auto compute = [&](auto tapc, const char* Ab, const char* Bb) {
    constexpr int tap = decltype(tapc)::value;
    constexpr int kh = tap / 3, kw = tap % 3;
    bf16x8 af[MI], bfr[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      af[mi] = *reinterpret_cast<const bf16x8*>(Ab + aofs[mi % RB][kw] + (kh + mi / RB) * HP * 64);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(Bb + bofs + ni * 1024);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)   // weights as A, pixels as B: the result tile is [channel][pixel] (see the epilogue)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
  };
  dma_halo(c_lo);
  static_for<NS - 1>([&](auto k) { dma_w(decltype(k)::value); });
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  int c = c_lo;
  for (; c + 1 < c_hi; ++c) {
    const char* Ab = ldsA + (c & 1) * A_BYTES;
    const int it0 = (c - c_lo) * 9;
    static_for<9>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap == 0) { if (!S2S_ABL(a.dbg & 4)) dma_halo(c + 1); else { static_for<HG>([&](auto) { dma16(g_zero_page, ldsA + ((c + 1) & 1) * A_BYTES + wave * 1024); }); } }
      if (!S2S_ABL(a.dbg & 1)) dma_w((it0 + tap + NS - 1) % NS);
      else if (!S2S_ABL(a.dbg & 512)) { static_for<BG>([&](auto) { dma16(g_zero_page, ldsB + wave * 1024); }); }
      if (!S2S_ABL(a.dbg & 2)) compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      if (!S2S_ABL(a.dbg & 256)) wait_vm<(NS - 2) * BG + (tap <= NS - 3 ? HG : 0)>();
      if (!S2S_ABL(a.dbg & 32)) __builtin_amdgcn_s_barrier();
#ifdef S2S_ABLATE
      // Timing only (S2S_CONV_DBG=1024): what a consumer-side BatchNorm + ReLU would cost this loop.
      // The input arrives by LDS-DMA, so the affine + ReLU has to be a read-modify-write pass over the next chunk's halo
      // image in LDS (it has landed by tap 2) plus a barrier, once per chunk.  scale / shift / floor are run-time values
      // that leave the data as it is (a real ReLU would zero half the operands and let the clock rise).
      if ((a.dbg & 1024) && tap == 8) {
        char* nb = ldsA + ((c + 1) & 1) * A_BYTES;
        const float sc = 1.0f + a.act_slope, sh = a.act_slope, fl = a.act_slope - 3.0e38f;
#pragma unroll
        for (int j = 0; j < HG; ++j) {
          bf16x8* p8 = reinterpret_cast<bf16x8*>(nb + (tid + 256 * j) * 16);
          bf16x8 v = *p8;
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (bf16_t)fmaxf(fmaf((float)v[k], sc, sh), fl);
          *p8 = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
#endif
    });
  }
  // the epilogue's per-channel bias: fetched here so that the load latency hides behind the last nine taps
  float biasr[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * WTN + ni * 16 + 4 * kp + j;
      biasr[ni][j] = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
    }
  {
    const char* Ab = ldsA + (c & 1) * A_BYTES;
    const int it0 = (c - c_lo) * 9;
    static_for<9>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap + NS - 1 < 9) dma_w((it0 + tap + NS - 1) % NS);
      compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      // slab tap + 1 must have landed; the slabs issued after it (up to the chunk's last one) may stay in flight
      constexpr int last_issued = (tap + NS - 1 < 9) ? tap + NS - 1 : 8;
      constexpr int younger = last_issued - (tap + 1) > 0 ? last_issued - (tap + 1) : 0;
      wait_vm<younger * BG>();
      __builtin_amdgcn_s_barrier();
    });
  }
  if (S2S_ABL(a.dbg & 16)) return;
  if (a.kpart) {
    float* const kpz = a.kpart + (long)blockIdx.z * ((long)a.B * a.H * a.W) * a.Cout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = wm * WTM + mi * 16 + cl;
      const int py = m / TW, px = m - py * TW;
      const int gy = y0 + py, gx = x0p + px;
      if (gy < a.H && gx < a.W) {
        float* const row = kpz + (((long)img * a.H + gy) * a.W + gx) * a.Cout;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int n = n0 + wn * WTN + ni * 16 + 4 * kp;
          if (n < a.Cout) *reinterpret_cast<f32x4*>(row + n) = acc[mi][ni];
        }
      }
    }
    return;
  }
  const OutMap om{1, 0, 0, a.H, a.W};
  if (a.direct_ep) {      // registers -> global memory (conv_epilogue16_direct), statistics in registers too
    if (a.stat_part) conv_epilogue16_direct<TH, TW, BN, WM, WN, false, true>(a, acc, biasr, img, y0, x0p, n0, tid, om, (long)tile_id * WM + wm);
    else if (a.ep_scale) conv_epilogue16_direct<TH, TW, BN, WM, WN, true>(a, acc, biasr, img, y0, x0p, n0, tid, om);
    else conv_epilogue16_direct<TH, TW, BN, WM, WN, false>(a, acc, biasr, img, y0, x0p, n0, tid, om);
  } else if (a.ep_scale) conv_epilogue16<TH, TW, BN, WM, WN, 2 * A_BYTES + NS * B_BYTES, true>(a, acc, biasr, smem, img, y0, x0p, n0, tid, tile_id, om);
  else conv_epilogue16<TH, TW, BN, WM, WN, 2 * A_BYTES + NS * B_BYTES, false>(a, acc, biasr, smem, img, y0, x0p, n0, tid, tile_id, om);
  if ((a.dbg & 64) && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) { g_clk[blockIdx.x * 4 + 0] = c0_; g_clk[blockIdx.x * 4 + 1] = __builtin_readcyclecounter(); g_clk[blockIdx.x * 4 + 2] = w0_; g_clk[blockIdx.x * 4 + 3] = wall_clock64(); }
}

// =========================================================================================================
// PERSISTENT form of conv3x3_dma16_kernel (round 3).  One workgroup per resident slot (2 per CU) walks a sequence of
// output tiles; the DMA streams never drain between tiles:
//   * the halo of the NEXT tile's first chunk is issued at tap 0 of the current tile's last chunk (into the halo buffer
//     the running chunk parity frees), and the weight ring simply continues into the next tile's first slabs, so a tile
//     starts on landed operands -- the per-tile prologue (index arithmetic, a DMA round trip behind `vmcnt(0)`, ~2.4 us
//     of every 12 us workgroup on the 64-channel layers at 256^2) is paid once per workgroup instead of once per tile;
//   * the epilogue is conv_epilogue16_direct: registers -> global memory, statistics in registers, no LDS, no barrier,
//     so it cannot collide with the prefetched halo / slabs, and its stores are simply counted into the next two taps'
//     `vmcnt` (they are younger than the slabs those taps wait for);
//   * no bias, no folded affine, no split-K: this kernel serves the training step (forward with BatchNorm statistics --
//     a conv bias in front of a train-mode BatchNorm cancels in the normalised output and is accounted for in the
//     running mean by s2s_bn_finalize -- and every data gradient); everything else stays on conv3x3_dma16_kernel.
// Tiles are dealt as before: linear id L -> (pixel tile, channel tile) through the XCD-aware cut (a.xsp x a.xsn), and
// workgroup w takes L = w, w + G, w + 2G, ... (G a multiple of 8: a workgroup stays inside its XCD's block).
// Every wave issues the same number of DMAs per tap whatever the tile (zero page for padding, ragged edges and "no next
// tile"), so all `s_waitcnt vmcnt(N)` are compile-time counts.
// =========================================================================================================
// STATS: 0 = none, 1 = one statistics row per (tile, wave row), 2 = carried over the workgroup's tiles (one row per
// (workgroup, wave row); launches with a single channel tile whose caller sized stat_part that way)
template <int TH, int TW, int BN, int WM, int WN, int NS, int STATS>
__global__ __launch_bounds__(256, 2) void conv3x3_pers16_kernel(Conv3x3Args a, int ntiles, int GX, int GY) {
  using T = bf16_t;
  constexpr int HP = TW + 4, HH_ = TH + 2, ROWS = HH_ * HP;
  constexpr int NGA = (ROWS + 15) / 16, HG = (NGA + 3) / 4;
  constexpr int A_BYTES = HG * 4 * 1024;
  constexpr int BG = BN / 64;
  constexpr int B_BYTES = BN * 64;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int RB = TW / 16;
  // VMEM stores a wave issues in one epilogue of a FULL tile: the 16-byte output stores, and with statistics the
  // 2 x 8 x NI/2 partial sums of a row's lane 0 (vmcnt counts stores too; partial tiles drain instead, see below)
  constexpr int NST = MI * (NI / 2) + (STATS == 1 ? (NI / 2) * 16 : 0);
  static_assert((NS - 2) * BG + HG + NST <= 63, "vmcnt is a 6-bit counter");
  static_assert(WM * WN == 4 && BN % 64 == 0 && NS == 4 && TW % 16 == 0 && WTM % TW == 0, "configuration");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem;
  char* const ldsB = smem + 2 * A_BYTES;
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);
  const char* __restrict__ wp = static_cast<const char*>(a.w);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int cl = lane & 15, kp = lane >> 4;
  const int ctot = a.c0 + a.c1;
  const int drow = lane >> 2, dslot = lane & 3;

  // ---- tile-invariant lane geometry ----
  // halo row / column / piece of this lane's DMA row j, packed: hy << 16 | hx << 8 | channel offset of the piece; rows of
  // the padding columns and past the image carry hy = 0x4000 (never inside an image)
  int ageo[HG];
#pragma unroll
  for (int j = 0; j < HG; ++j) {
    const int row = (wave + 4 * j) * 16 + drow;
    const int hy = row / HP, hx = row - hy * HP;
    ageo[j] = (((row < ROWS && hx < TW + 2) ? hy : 0x4000) << 16) | (hx << 8) | ((dslot ^ (((hx >> 2) & 1) << 1)) * 8);
  }
  int aofs[RB][3];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int px = rb * 16 + cl + kw;
      aofs[rb][kw] = ((wm * (WTM / TW)) * HP + px) * 64 + ((kp ^ (((px >> 2) & 1) << 1)) << 4);
    }
  const int nrow = wn * WTN + cl;
  const int bofs = nrow * 64 + ((kp ^ ((-(nrow >> 2)) & 3)) << 4);
  int wrow_off[BG];                                    // byte offset of this lane's weight row piece inside a slab, tile n0 = 0
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int n = (wave + 4 * j) * 16 + drow;
    wrow_off[j] = (n * 32 + ((dslot ^ ((-(n >> 2)) & 3)) * 8)) * 2;
  }

  // ---- per-tile state ----
  struct Tile { int img, y0, x0p, n0, id; };
  auto locate = [&](int L) {
    int bt, by;
    if (a.xsp) {
      const unsigned xcd = (unsigned)L & 7u, k = (unsigned)L >> 3;
      const unsigned xp = xcd % (unsigned)a.xsp, xn = xcd / (unsigned)a.xsp;
      const unsigned P = (unsigned)GX / (unsigned)a.xsp, Nn = (unsigned)GY / (unsigned)a.xsn;
      const unsigned kp_ = k / Nn, kn_ = k - kp_ * Nn;
      bt = (int)(xp * P + kp_); by = (int)(xn * Nn + kn_);
    } else {
      by = L / GX; bt = L - by * GX;
    }
    Tile t;
    t.id = bt;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    t.img = bt / a.tilesY;
    t.y0 = ty * TH; t.x0p = tx * TW; t.n0 = by * BN;
    return t;
  };
  // weight stream: pointer to the next slab piece this lane issues, slabs left in the tile it belongs to
  const char* wq[BG];
  int wqstep[BG];
  auto wq_start = [&](int n0, bool live) {
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      const int n = (wave + 4 * j) * 16 + drow;
      const bool ok = live && n0 + n < a.Cout;
      wq[j] = ok ? wp + (long)n0 * 64 + wrow_off[j] : g_zero_page;
      wqstep[j] = ok ? a.Cout * 64 : 0;
    }
  };
  auto dma_w = [&](int slot) {
    char* dst = ldsB + slot * B_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      dma16(wq[j], dst + j * 4096);
      wq[j] += wqstep[j];
    }
  };
  // Halo DMA.  The source of a chunk is uniform (the launcher sends c0 % 32 != 0 to the other kernel): one base pointer
  // in scalar registers plus a 32-bit byte offset per lane (inputs are < 4 GB), instead of a 64-bit address per piece.
  auto halo_pix = [&](const Tile& t, bool live, int (&apix)[HG]) {
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      const int gy = t.y0 - 1 + (ageo[j] >> 16), gx = t.x0p - 1 + ((ageo[j] >> 8) & 0xff);
      apix[j] = (live && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (t.img * a.H + gy) * a.W + gx : -1;
    }
  };
  auto dma_halo = [&](const int (&apix)[HG], int c, int buf) {
    char* dst = ldsA + buf * A_BYTES + wave * 1024;
    const bool second = c * 32 >= a.c0;
    const char* const base = reinterpret_cast<const char*>(second ? x1 : x0);
    const int ld = second ? a.ld1 : a.ld0, ch0 = c * 32 - (second ? a.c0 : 0);
    const bool inrange = c * 32 < ctot;                 // (a chunk past the channels: only when ctot % 32 != 0 rounds up)
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      const int ch = ch0 + (ageo[j] & 0xff);
      const unsigned off = ((unsigned)apix[j] * (unsigned)ld + (unsigned)ch) * 2u;
      const bool ok = apix[j] >= 0 && inrange && c * 32 + (ageo[j] & 0xff) < ctot;
      const void* g = ok ? static_cast<const void*>(base + off) : static_cast<const void*>(g_zero_page);
      dma16(g, dst + j * 4096);
    }
  };

  f32x4 acc[MI][NI];
  auto compute = [&](auto tapc, const char* Ab, const char* Bb) {
    constexpr int tap = decltype(tapc)::value;
    constexpr int kh = tap / 3, kw = tap % 3;
    bf16x8 af[MI], bfr[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      af[mi] = *reinterpret_cast<const bf16x8*>(Ab + aofs[mi % RB][kw] + (kh + mi / RB) * HP * 64);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(Bb + bofs + ni * 1024);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
  };

  // ---- prologue: first tile's chunk-0 halo and first NS - 1 slabs ----
  const int G = gridDim.x;
  int L = blockIdx.x;
  Tile cur = locate(L);
  bool has_next = L + G < ntiles;
  Tile nxt = cur;
  if (has_next) nxt = locate(L + G);
  int apix_cur[HG];
  halo_pix(cur, true, apix_cur);
  wq_start(cur.n0, true);
  dma_halo(apix_cur, 0, 0);
  static_for<NS - 1>([&](auto k) { dma_w(decltype(k)::value); });
  wait_vm<0>();                                        // (the first tile relies on this drain: see the wait counts below)
  __builtin_amdgcn_s_barrier();

  int gc = 0;                                          // chunks this workgroup has run: halo buffer = gc & 1, slab = 9 gc + tap
  // one chunk = nine taps.  FIRST: the tile's first chunk -- its taps 0 and 1 wait for slabs that were issued BEFORE the
  // previous tile's epilogue, so that epilogue's NST stores are younger and are counted in.  (For the workgroup's very
  // first tile there are no such stores; the larger count is then merely permissive, and safe, because the prologue's
  // vmcnt(0) has already landed slabs 1 and 2.)
  auto chunk = [&](auto firstc, int c) {
    constexpr bool FIRST = decltype(firstc)::value;
    const char* Ab = ldsA + (gc & 1) * A_BYTES;
    const int it0 = gc * 9;
    const bool last = c + 1 == a.nchunk;
    static_for<9>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap == 0) {                                  // the next chunk in sequence: this tile's, or the next tile's first
        if (!last) dma_halo(apix_cur, c + 1, (gc + 1) & 1);
        else {
          int apix_nxt[HG];
          halo_pix(nxt, has_next, apix_nxt);
          dma_halo(apix_nxt, 0, (gc + 1) & 1);
        }
      }
      // the ring runs NS - 1 slabs ahead: at tap 9 - (NS - 1) of a tile's last chunk it crosses into the next tile's
      if (tap == 9 - (NS - 1) && last) wq_start(nxt.n0, has_next);
      dma_w((it0 + tap + NS - 1) % NS);
      compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      // slab it + 1 must have landed.  Younger than it: the NS - 2 slabs issued since, the halo of tap 0 while tap <=
      // NS - 3, and in taps 0 / 1 of a tile's first chunk the previous epilogue's stores
      wait_vm<(NS - 2) * BG + (tap <= NS - 3 ? HG : 0) + ((FIRST && tap <= 1) ? NST : 0)>();
      __builtin_amdgcn_s_barrier();
    });
    ++gc;
  };
  // STATS == 2 (one channel tile, the caller sized stat_part for workgroups x WM rows): the statistics of a wave row
  // stay in registers over all tiles of the workgroup -- per tile only the 2 x 64 accumulations remain, the DPP fold and
  // the 32 single-lane stores (which the next tile's third tap had to see retired: vmcnt is in order) happen once
  float cs1[STATS == 2 ? NI / 2 : 1][8], cs2[STATS == 2 ? NI / 2 : 1][8];
  if constexpr (STATS == 2) {
#pragma unroll
    for (int pr = 0; pr < NI / 2; ++pr)
#pragma unroll
      for (int k = 0; k < 8; ++k) { cs1[pr][k] = 0.f; cs2[pr][k] = 0.f; }
  }
  for (;;) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mi][ni][j] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    chunk(std::true_type{}, 0);
    for (int c = 1; c < a.nchunk; ++c) chunk(std::false_type{}, c);
    {
      const OutMap om{1, 0, 0, a.H, a.W};
      const float nobias[NI][4] = {};
      if constexpr (STATS == 2)
        conv_epilogue16_direct<TH, TW, BN, WM, WN, false, true, true>(a, acc, nobias, cur.img, cur.y0, cur.x0p, cur.n0, tid, om, -1, cs1, cs2);
      else
        conv_epilogue16_direct<TH, TW, BN, WM, WN, false, STATS == 1>(a, acc, nobias, cur.img, cur.y0, cur.x0p, cur.n0, tid, om, (long)cur.id * WM + wm);
      // a partial tile (image edge, last channel tile) may skip store instructions whose lanes are all out of range, so
      // its store count is not the compile-time one: drain instead (everything older has then landed, and the larger
      // counts of the next two taps are merely permissive)
      if (!(cur.y0 + TH <= a.H && cur.x0p + TW <= a.W && cur.n0 + BN <= a.Cout)) wait_vm<0>();
    }
    if (!has_next) break;
    L += G;
    cur = nxt;
    halo_pix(cur, true, apix_cur);
    has_next = L + G < ntiles;
    if (has_next) nxt = locate(L + G);
  }
  if constexpr (STATS == 2) conv_stats_flush<BN, WN>(a, cs1, cs2, cur.n0, tid, (long)blockIdx.x * WM + wm);
}

// Grid and XCD cut of a persistent launch over GX pixel tiles x GY channel tiles (shared with s2s_conv3x3_stat_rows: with
// one channel tile the statistics take grid * WM rows)
struct PersPlan { int grid, xsp, xsn; };
inline PersPlan pers_plan(const Conv3x3Args& a, int GX, int GY) {
  PersPlan p{1, 0, 1};
  const long ntiles = (long)GX * GY;
  if (ntiles % 8 == 0) {
    const double act = 2.0 * a.B * a.H * a.W * (a.c0 + a.c1), wb = 2.0 * 9 * (a.c0 + a.c1) * a.Cout;
    double best = 0;
    for (int sp = 8; sp >= 1; sp >>= 1) {
      const int sn = 8 / sp;
      if (GX % sp || GY % sn) continue;
      const double cost = sn * act + sp * wb;
      if (!p.xsp || cost < best) { best = cost; p.xsp = sp; p.xsn = sn; }
    }
  }
  constexpr int slots = 512;                           // 2 per CU (swept in round 3: no better grid)
  int grid = (int)(ntiles < slots ? ntiles : slots);
  if (p.xsp && grid > 8) grid -= grid % 8;             // a stride that is a multiple of 8 keeps a workgroup inside its XCD's block
  p.grid = grid < 1 ? 1 : grid;
  return p;
}

template <int TH, int TW, int BN, int WM, int WN>
int launch_pers16(Conv3x3Args& a, hipStream_t s) {
  constexpr int NS = 4;
  constexpr int ROWS = (TH + 2) * (TW + 4);
  constexpr int HG = ((ROWS + 15) / 16 + 3) / 4;
  constexpr int lds = 2 * HG * 4 * 1024 + NS * BN * 64;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  a.tilesY = cdiv(a.H, TH);
  a.tilesX = cdiv(a.W, TW);
  auto kern_s = conv3x3_pers16_kernel<TH, TW, BN, WM, WN, NS, 1>;
  auto kern_c = conv3x3_pers16_kernel<TH, TW, BN, WM, WN, NS, 2>;
  auto kern_d = conv3x3_pers16_kernel<TH, TW, BN, WM, WN, NS, 0>;
  static unsigned long long attr_s = 0, attr_c = 0, attr_d = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern_s), lds, &attr_s)) return rc;
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern_c), lds, &attr_c)) return rc;
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern_d), lds, &attr_d)) return rc;
  const int GX = a.B * a.tilesY * a.tilesX, GY = cdiv(a.Cout, BN);
  const long ntiles = (long)GX * GY;
  const PersPlan pl = pers_plan(a, GX, GY);
  a.xsp = pl.xsp; a.xsn = pl.xsn;
  const int grid = pl.grid;
  // statistics rows: one per (workgroup, wave row) when the caller sized the buffer for that (s2s_conv3x3_stat_rows) and
  // there is a single channel tile; else one per (tile, wave row)
  a.stat_carry = a.stat_part && GY == 1 && a.stat_rows_req == (long)grid * WM;
  if (a.stat_part && !a.stat_carry && a.stat_rows_req && a.stat_rows_req != (long)GX * WM) return S2S_ERR_SHAPE;
  a.stat_rows = a.stat_carry ? (long)grid * WM : (long)GX * WM;
  if (a.stat_carry) hipLaunchKernelGGL(kern_c, dim3(grid), dim3(256), lds, s, a, (int)ntiles, GX, GY);
  else if (a.stat_part) hipLaunchKernelGGL(kern_s, dim3(grid), dim3(256), lds, s, a, (int)ntiles, GX, GY);
  else hipLaunchKernelGGL(kern_d, dim3(grid), dim3(256), lds, s, a, (int)ntiles, GX, GY);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// =========================================================================================================
// STAGED persistent kernel (round 4): one workgroup of 8 waves per CU walks 16 x 32-pixel x 64-channel tiles with ONE
// barrier per 32-channel chunk (nine taps) instead of one per tap.
//   * LDS = two stages of { halo of one chunk (18 x 36 rows x 64 B = 41 KB) | the chunk's nine weight slabs (36 KB) } =
//     154 KB.  At the top of a step the workgroup waits for the stage's DMAs (issued one step earlier: a whole chunk,
//     ~2 us, to land), passes the barrier, issues the NEXT step's DMAs into the other stage -- this tile's next chunk, or
//     the next tile's first -- and runs nine taps without further synchronisation.
//   * inside a chunk the fragments are double-buffered by tap parity: the ds_read_b128 of tap t + 1 are issued one per
//     MFMA behind the first MFMAs of tap t (left to the compiler the loop waited `lgkmcnt(0)` right behind freshly issued
//     reads several times per tap).  With operands in LDS this loop runs at ~90 % of the MFMA rate the clock allows
//     (64 -> 64 at 256^2 without DMAs and epilogue: 41 us for 77 GFLOP, profiles/r04_stage_ablation.txt).
//   * RESIDENT (<= 64 input channels: 64 -> 64 at 256^2 forward and data gradient, the 64 -> 192 data gradient, the stem
//     on its 8-channel NHWC image): the two stages' slabs ARE the whole filter of a channel tile (<= 72 KB): they are loaded once and stay; only halos stream, and
//     jobs are ordered channel tile slowest so that a workgroup reloads the filter at most GY - 1 times.  The per-tap
//     kernels above re-fetch those 72 KB for every 256-pixel tile.
//   * streaming (any other channel count): jobs are ordered channel tile fastest inside an XCD's block of pixel tiles, so
//     the GY workgroups that share a pixel tile run at the same time and its halo comes from the XCD's L2.
//   * DEFER_EP (launches without statistics): a tile's accumulators are rounded and exchanged right after its last tap
//     (32 registers), and the 16-byte stores are issued one per tap behind the MFMAs of the next tile's first chunk: all
//     waves pass the barriers together, so an epilogue in place leaves the matrix pipe idle on every SIMD at once.
//   * STATS = 2: BatchNorm partial sums carried in registers over the workgroup's tiles of one channel tile, written as row
//     (workgroup, wave row) when the channel tile changes and at the end; channel tiles a workgroup never visited get
//     zero rows.  stat_part is [2][Cout][workgroups x 8].
// =========================================================================================================
template <int STATS, bool DEFER_EP, bool RESIDENT, bool AFFINE = false>
__global__ __launch_bounds__(512, 1) void conv3x3_stage_kernel(Conv3x3Args a, int njobs, int GX, int GY) {
  using T = bf16_t;
  constexpr int TH = 16, TW = 32, BN = 64, WM = 8, WN = 1;
  constexpr int HP = TW + 4, ROWS = (TH + 2) * HP;
  constexpr int NGA = (ROWS + 15) / 16, HG = (NGA + WM - 1) / WM;
  constexpr int A_BYTES = NGA * 1024;
  constexpr int B_BYTES = BN * 64, NGB = 9 * 4, BGW = (NGB + WM - 1) / WM;   // 16-row weight groups of a chunk / per wave
  constexpr int STAGE = A_BYTES + 9 * B_BYTES;
  constexpr int WTM = TH * TW / WM, MI = WTM / 16, NI = BN / 16, RB = TW / 16;
  constexpr int NST = MI * (NI / 2);
  static_assert(NST <= 8, "one deferred store unit per tap");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const T* __restrict__ x1 = static_cast<const T*>(a.x1);
  const char* __restrict__ wp = static_cast<const char*>(a.w);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave;
  const int cl = lane & 15, kp = lane >> 4;
  const int drow = lane >> 2, dslot = lane & 3;
  const int ctot = a.c0 + a.c1;

  // halo row / column / piece of this lane's DMA row j (as in conv3x3_pers16_kernel; group = wave + 8 j, groups past the
  // halo are not issued)
  int ageo[HG];
#pragma unroll
  for (int j = 0; j < HG; ++j) {
    const int row = (wave + WM * j) * 16 + drow;
    const int hy = row / HP, hx = row - hy * HP;
    ageo[j] = (((row < ROWS && hx < TW + 2) ? hy : 0x4000) << 16) | (hx << 8) | ((dslot ^ (((hx >> 2) & 1) << 1)) * 8);
  }
  int aofs[RB][3];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int px = rb * 16 + cl + kw;
      aofs[rb][kw] = ((wm * (WTM / TW)) * HP + px) * 64 + ((kp ^ (((px >> 2) & 1) << 1)) << 4);
    }
  int bofs = cl * 64 + ((kp ^ ((-(cl >> 2)) & 3)) << 4);

  struct Tile { int img, y0, x0p, n0; };
  auto locate = [&](int L) {
    int bt, by;
    if (a.xsp) {                                       // XCD x = L & 7 owns pixel tiles [x P, (x + 1) P), all channel tiles
      const unsigned xcd = (unsigned)L & 7u, k = (unsigned)L >> 3, P = (unsigned)GX >> 3;
      if constexpr (RESIDENT) { by = (int)(k / P); bt = (int)(xcd * P + (k - (unsigned)by * P)); }
      else { const unsigned kq = k / (unsigned)GY; by = (int)(k - kq * (unsigned)GY); bt = (int)(xcd * P + kq); }
    } else if constexpr (RESIDENT) {
      by = L / GX; bt = L - by * GX;
    } else {
      bt = L / GY; by = L - bt * GY;
    }
    Tile t;
    const int tx = bt % a.tilesX; bt /= a.tilesX;
    const int ty = bt % a.tilesY;
    t.img = bt / a.tilesY;
    t.y0 = ty * TH; t.x0p = tx * TW; t.n0 = by * BN;
    return t;
  };
  // Operand DMAs: buffer_load_dwordx4 ... lds through raw buffer resources -- a scalar base, a 32-bit byte offset per
  // lane, and ZERO FILL for offsets past num_records: padding pixels, rows past Cout and channel pieces past the input
  // take the offset OOB, so there is no zero page, no 64-bit address per lane and no pointer select (scripts/probe/
  // buffer_lds_probe.hip checks the destination layout and the zero fill on the device).
  constexpr unsigned OOB = 0xffffff00u;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wp), 0, (int)((long)a.nchunk * 9 * a.Cout * 64), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x0), 0, (int)((((long)a.B * a.H * a.W - 1) * a.ld0 + a.c0) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.c1 ? x1 : x0), 0, a.c1 ? (int)((((long)a.B * a.H * a.W - 1) * a.ld1 + a.c1) * 2) : 0, 0x00020000);
  // the nine slabs of chunk c, channel tile n0, into stage `st` (rows swizzled as the kernels above).  Group g = wave + 8 j
  // is rows (g & 3) * 16 + drow of slab g >> 2: the row is the same for every j, only the slab (a scalar) moves
  const int wrow = (wave & 3) * 16 + drow;
  const unsigned wsw = (unsigned)wrow * 64u + (unsigned)((dslot ^ ((-(wrow >> 2)) & 3)) << 4);
  auto dma_weights = [&](int n0, int c, int st) {
    char* dst = smem + st * STAGE + A_BYTES + wave * 1024;
    const unsigned voff = n0 + wrow < a.Cout ? wsw : OOB;
#pragma unroll
    for (int j = 0; j < BGW; ++j) {
      if (wave + WM * j >= NGB || S2S_ABL(a.dbg & 1)) continue;   // (wave-uniform)
      const int tap = (wave >> 2) + (WM / 4) * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(dst + j * (WM * 1024)), 16, voff,
                                               ((c * 9 + tap) * a.Cout + n0) * 64, 0, 0);
    }
  };
  auto halo_pix = [&](const Tile& t, int (&apix)[HG]) {
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      const int gy = t.y0 - 1 + (ageo[j] >> 16), gx = t.x0p - 1 + ((ageo[j] >> 8) & 0xff);
      apix[j] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (t.img * a.H + gy) * a.W + gx : -1;
    }
  };
  auto dma_halo = [&](const int (&apix)[HG], int c, int st) {
    char* dst = smem + st * STAGE + wave * 1024;
    const bool second = c * 32 >= a.c0;
    const int ld2 = (second ? a.ld1 : a.ld0) * 2, ch0 = c * 32 - (second ? a.c0 : 0);
    const int left = ctot - c * 32;                    // channels of this chunk that exist (< 32 only when ctot % 32 != 0)
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      if (wave + WM * j >= NGA || S2S_ABL(a.dbg & 4)) continue;   // (wave-uniform)
      const int pc = ageo[j] & 0xff;
      const unsigned voff = (apix[j] >= 0 && pc < left) ? (unsigned)apix[j] * (unsigned)ld2 + (unsigned)(pc * 2) : OOB;
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_void_t*)(dst + j * (WM * 1024)), 16, voff, ch0 * 2, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_void_t*)(dst + j * (WM * 1024)), 16, voff, ch0 * 2, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
  // (aofs / bofs are the CURRENT stage's byte offsets into smem: step() moves them by +-STAGE, everything else in a
  //  fragment address is an instruction immediate)
  auto afrag = [&](auto tapc, int mi) {
    constexpr int tap = decltype(tapc)::value;
    constexpr int kh = tap / 3, kw = tap % 3;
    return *reinterpret_cast<const bf16x8*>(smem + aofs[mi % RB][kw] + (kh + mi / RB) * HP * 64);
  };
  auto bfrag = [&](auto tapc, int ni) {
    constexpr int tap = decltype(tapc)::value;
    return *reinterpret_cast<const bf16x8*>(smem + bofs + (A_BYTES + tap * B_BYTES + ni * 1024));
  };

  // ---- epilogue: rounded + exchanged values (v_permlane16_swap, as conv_epilogue16_direct), then one 16-byte store per
  // unit u = (mi, pr): this lane's pixel of 16-pixel block mi, eight channels of channel pair pr ----
  T* __restrict__ yout = static_cast<T*>(a.y);
  uint4 outv[MI][NI / 2];
  float cs1[STATS == 2 ? NI / 2 : 1][8], cs2[STATS == 2 ? NI / 2 : 1][8];
  auto stats_zero = [&]() {
    if constexpr (STATS == 2) {
#pragma unroll
      for (int pr = 0; pr < NI / 2; ++pr)
#pragma unroll
        for (int k = 0; k < 8; ++k) { cs1[pr][k] = 0.f; cs2[pr][k] = 0.f; }
    }
  };
  stats_zero();
  // AFFINE (the eval-mode forward of the sampler: conv bias + folded BatchNorm + ReLU): y = max(acc * esc + esh', 0) with
  // esh' = bias * esc + esh, this lane's sixteen channels of the current channel tile kept in registers
  float esc[AFFINE ? NI : 1][4], esh[AFFINE ? NI : 1][4];
  auto load_affine = [&](int n0) {
    if constexpr (AFFINE) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + ni * 16 + 4 * kp;           // (Cout % 8 == 0: four channels are in or out together)
        const bool ok = n < a.Cout;
        const f32x4 sc = ok ? *reinterpret_cast<const f32x4*>(a.ep_scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = ok ? *reinterpret_cast<const f32x4*>(a.ep_shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 bi = (ok && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) { esc[ni][j] = sc[j]; esh[ni][j] = fmaf(bi[j], sc[j], sh[j]); }
      }
    }
  };
  auto pack = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int pr = 0; pr < NI / 2; ++pr) {
        unsigned w[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          bf16x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = acc[mi][2 * pr + b][j];
            if constexpr (AFFINE) {
              t = fmaf(t, esc[2 * pr + b][j], esh[2 * pr + b][j]);
              if (a.relu) t = fmaxf(t, 0.f);
            }
            pk[j] = (bf16_t)t;
          }
          const uint2 u = __builtin_bit_cast(uint2, pk);
          w[b][0] = u.x; w[b][1] = u.y;
        }
        const auto r0 = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
        outv[mi][pr] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      }
  };
  // Stores go through a raw buffer resource as well: the tile / unit part of the address is a scalar offset, the lane's
  // part (pixel column cl, its eight channels) ONE tile-invariant register, and lanes outside the image or past Cout take
  // the offset out of range -- the hardware drops them (scripts/probe/buffer_store_probe.hip), so a store unit has no
  // branch and the chunk stays one scheduling region.
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(yout, 0, (int)((((long)a.B * a.H * a.W - 1) * a.ldy + a.Cout) * 2), 0x00020000);
  const int nl = (kp & 1) * 16 + (kp >> 1) * 8;
  const unsigned vst = (unsigned)(cl * a.ldy + nl) * 2u;
  auto store_unit = [&](auto uc, const Tile& t) {
    constexpr int u = decltype(uc)::value, mi = u / (NI / 2), pr = u % (NI / 2);
    const int gy = t.y0 + wm * (WTM / TW) + mi / RB, gx0 = t.x0p + (mi % RB) * 16, nb = t.n0 + pr * 32;   // (scalars)
    const bool ok = gy < a.H && gx0 + cl < a.W && nb + nl < a.Cout;
    if constexpr (STATS == 2) {
      const bf16x8 val = __builtin_bit_cast(bf16x8, outv[mi][pr]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = ok ? (float)val[k] : 0.f;
        cs1[pr][k] += v;
        cs2[pr][k] = fmaf(v, v, cs2[pr][k]);
      }
    }
    if (!S2S_ABL(a.dbg & 8))
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, outv[mi][pr]), rs_y, ok ? vst : OOB,
                                             (((t.img * a.H + gy) * a.W + gx0) * a.ldy + nb) * 2, 0);
  };

  // One chunk = nine taps on stage `st`.  DEFER: the previous tile's store units ride behind taps 0 .. NST - 1.
  auto chunk = [&](auto deferc, const Tile& prv) {
    constexpr bool DEFER = decltype(deferc)::value;
    // Fragments: the pixel (A) side double-buffered by tap parity, the weight (B) side in ONE set -- the MFMAs run channel
    // block by channel block, and block ni's fragment of the next tap is read as soon as its four MFMAs are issued (it is
    // needed twelve MFMAs later).  Per tap: four A reads behind the first four MFMAs, then one B read per channel block.
    bf16x8 af[2][MI], bfr[NI];
    using T0 = std::integral_constant<int, 0>;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[ni] = bfrag(T0{}, ni);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[0][mi] = afrag(T0{}, mi);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);   // (these reads, not the next tap's, fill the first group)
    static_for<9>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value, cb = tap & 1;
      using TN = std::integral_constant<int, (tap < 8 ? tap + 1 : 8)>;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[cb][mi], acc[mi][ni], 0, 0, 0);
          if (tap < 8 && ni == 0) af[cb ^ 1][mi] = afrag(TN{}, mi);
        }
        if (tap < 8) bfr[ni] = bfrag(TN{}, ni);
      }
      if constexpr (DEFER && tap < NST) store_unit(std::integral_constant<int, (tap < NST ? tap : 0)>{}, prv);
      if constexpr (tap < 8) {
#pragma unroll
        for (int k = 0; k < MI; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int k = 1; k < NI; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, MI, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    });
  };

  // Phase offset: the 256 workgroups have identical work and start together, so without it the whole chip alternates
  // between "every CU issues its DMAs" and "every CU runs MFMAs" -- operand traffic and matrix work do not overlap ACROSS
  // CUs, and the 256^2 layers this kernel serves are the HBM-heaviest of the step.  Workgroup w of an XCD sleeps
  // (w mod 16) x 0.66 us before its first DMA: 16 phases over ~10 us, about one tile period of 64 -> 64.  Measured on the
  // sustained CFM step (1000 steps; profiles/r04_stage_ab.txt): one box 7.66 ms without, 7.62 / 7.45 / 7.45 ms with 0.22 /
  // 0.44 / 0.66 us per phase (per-tap kernels throughout: 7.65 ms); another 7.16 ms at 0.66 against 7.20-7.33 at 0.44 /
  // 0.88 / 1.3 / 1.8 and 7.41 per-tap -- the offset costs the last phase's delay once per launch and returns several
  // times that.  (The same offset on the first round of conv3x3_dma16_kernel's
  // workgroups LOSES monotonically -- 7.09 / 7.14 / 7.21 / 7.30 ms for 0 / 0.22 / 0.44 / 0.66 us per phase: those layers are
  // not operand-traffic-bound -- and on conv3x3_wgrad_dma_kernel it does nothing.)
  for (int i = 0, d = 3 * (int)((blockIdx.x >> 3) & 15); i < d; ++i) __builtin_amdgcn_s_sleep(8);
  const int G = gridDim.x;
  int L = blockIdx.x;
  Tile cur = locate(L);
  bool has_next = L + G < njobs;
  Tile nxt = cur, prv = cur;
  if (has_next) nxt = locate(L + G);
  int apix[HG];                                        // halo pixels of the tile whose DMAs are issued next
  halo_pix(cur, apix);
  dma_halo(apix, 0, 0);
  dma_weights(cur.n0, 0, 0);
  if constexpr (RESIDENT) dma_weights(cur.n0, a.nchunk > 1 ? 1 : 0, 1);   // (one chunk: both stages hold it)
  load_affine(cur.n0);
  bool pend = false, pend_full = false;                // a packed tile waits in outv / it was a full tile (NST stores)
  bool ynst = false;                                   // exactly NST stores are younger than the newest DMA
  int st = 0;
  const int by_first = cur.n0 / BN;

  auto step = [&](int c, auto mayc) {
    constexpr bool MAY_DEFER = decltype(mayc)::value;
    // this step's operands were issued one step ago; younger than them are at most the NST stores of a full tile
    if (ynst) wait_vm<NST>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                      // ... landed for every wave, and everyone is past the previous step
    ynst = false;
    if (c + 1 < a.nchunk) {
      dma_halo(apix, c + 1, st ^ 1);
      if constexpr (!RESIDENT) dma_weights(cur.n0, c + 1, st ^ 1);
    } else if (has_next) {                             // (this tile's halos are all issued: apix moves on to the next tile)
      halo_pix(nxt, apix);
      dma_halo(apix, 0, st ^ 1);
      if constexpr (!RESIDENT) dma_weights(nxt.n0, 0, st ^ 1);
    }
    if (S2S_ABL(a.dbg & 2)) {
    } else if (MAY_DEFER && pend) {
      chunk(std::true_type{}, prv);
      ynst = pend_full;
      pend = false;
    } else {
      chunk(std::false_type{}, prv);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const int d = st ? -STAGE : STAGE;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) aofs[rb][kw] += d;
      bofs += d;
    }
    st ^= 1;
  };
  for (;;) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mi][ni][j] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    // (the first chunk apart: the deferred stores ride only there, so `outv` is dead in the loop over the others)
    step(0, std::integral_constant<bool, DEFER_EP>{});
    for (int c = 1; c < a.nchunk; ++c) step(c, std::false_type{});
    if (!S2S_ABL(a.dbg & 16)) {
      pack();
      prv = cur;
      const bool full = cur.y0 + TH <= a.H && cur.x0p + TW <= a.W && cur.n0 + BN <= a.Cout;
      if constexpr (DEFER_EP) {
        pend = true;
        pend_full = full;
      } else {
        static_for<NST>([&](auto uc) { store_unit(uc, prv); });
        ynst = full;
      }
    }
    if (!has_next) break;
    if (nxt.n0 != cur.n0) {
      load_affine(nxt.n0);                             // (pack() of the current tile is done)
      if constexpr (STATS == 2) {                      // (a wave row's sums of this channel tile: written once)
        conv_stats_flush<BN, WN>(a, cs1, cs2, cur.n0, tid, (long)blockIdx.x * WM + wm);
        stats_zero();
        ynst = false;
      }
      if constexpr (RESIDENT) {                        // everyone is done with the filter: load the next channel tile's
        __builtin_amdgcn_s_barrier();
        dma_weights(nxt.n0, 0, 0);
        dma_weights(nxt.n0, a.nchunk > 1 ? 1 : 0, 1);
        ynst = false;
      }
    }
    L += G;
    cur = nxt;
    has_next = L + G < njobs;
    if (has_next) nxt = locate(L + G);
  }
  if (pend) static_for<NST>([&](auto uc) { store_unit(uc, prv); });
  if constexpr (STATS == 2) {
    conv_stats_flush<BN, WN>(a, cs1, cs2, cur.n0, tid, (long)blockIdx.x * WM + wm);
    // zero rows for the channel tiles this workgroup never visited: RESIDENT job lists are monotonic in the channel tile;
    // streaming ones keep ONE channel tile per workgroup (stage_grid() admits statistics only when GY divides 32)
    stats_zero();
    const int by_last = cur.n0 / BN;
    for (int by = 0; by < GY; ++by) {
      if (by < by_first || by > by_last) conv_stats_flush<BN, WN>(a, cs1, cs2, by * BN, tid, (long)blockIdx.x * WM + wm);
    }
  }
}

// Does this bf16 launch run on the staged kernel, and on how many workgroups?  (dispatch() and s2s_conv3x3_stat_rows()
// share the decision: a statistics launch writes grid x 8 rows.)  0 = no.
inline int stage_grid(const Conv3x3Args& a, bool stats) {
  static const int on = [] { const char* e = getenv("S2S_CONV_STAGE"); return e ? atoi(e) : 1; }();
  // (a folded affine -- the eval-mode forward -- has its own instantiation; a bias alone or statistics beside an affine do not)
  if (!on || (a.c1 && a.c0 % 32) || (a.bias && !a.ep_scale) || (a.ep_scale && stats) || a.kpart || a.act || a.y2 ||
      !a.direct_ep || (a.dbg & 64)) return 0;
  if ((double)a.B * a.H * a.W * a.ld0 * 2 >= 4.0e9 || (double)a.B * a.H * a.W * (a.c1 ? a.ld1 : 0) * 2 >= 4.0e9 ||
      (double)a.B * a.H * a.W * a.ldy * 2 >= 2.0e9) return 0;          // 32-bit byte offsets into each tensor (the store's
                                                                         // scalar offset is formed in signed arithmetic)
  const long GX = (long)a.B * cdiv(a.H, 16) * cdiv(a.W, 32), GY = cdiv(a.Cout, 64);
  const long njobs = GX * GY;
  constexpr int slots = 256;                           // one workgroup per CU (154 KB of LDS)
  if (njobs < 2L * slots || njobs > 0x7fffffffL) return 0;             // a walk of one tile buys nothing
  // Measured against the per-tap kernels on the production shapes (batch 16, scripts/conv_bench.py, round 4): the resident
  // form wins everywhere it applies (64 -> 64 at 256^2: 108 -> 83 us, the 64 -> 192 data gradient 283 -> 215 us); the
  // streaming form wins with ONE channel tile (192 -> 64 at 256^2: 232 -> 211 us, the 128 -> 64 data gradient 52 -> 50 us)
  // and ties or loses by up to 15 % with more (128 -> 128 at 128^2: 81 = 81 us, 256 -> 256 at 64^2: 69 -> 78 us) -- those
  // stay on the per-tap kernels.  S2S_CONV_STAGE=2 admits them too (the parity tests do, to cover the order and the
  // statistics rows of GY > 1).
  if (a.c0 + a.c1 > 64 && GY > 1 && on < 2) return 0;
  // statistics are carried per (workgroup, channel tile): the streaming order keeps one channel tile per workgroup only
  // when GY divides the 32 workgroups of an XCD
  if (stats && a.c0 + a.c1 > 64 && (GY > 32 || (32 % GY))) return 0;
  return slots;
}

inline int launch_stage(Conv3x3Args& a, int grid, hipStream_t s) {
  constexpr int lds = 2 * (41 * 1024 + 9 * 4096);
  static_assert(lds <= 160 * 1024, "LDS budget");
  a.tilesY = cdiv(a.H, 16);
  a.tilesX = cdiv(a.W, 32);
  const bool resident = a.c0 + a.c1 <= 64;             // one or two chunks: the whole filter fits the two stages
  const int form = a.stat_part ? 1 : (a.ep_scale ? 2 : 0);
  auto kern = form == 1 ? (resident ? conv3x3_stage_kernel<2, false, true> : conv3x3_stage_kernel<2, false, false>)
            : form == 2 ? (resident ? conv3x3_stage_kernel<0, false, true, true> : conv3x3_stage_kernel<0, false, false, true>)
                        : (resident ? conv3x3_stage_kernel<0, true, true> : conv3x3_stage_kernel<0, true, false>);
  static unsigned long long attr[6] = {0, 0, 0, 0, 0, 0};
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr[form * 2 + (resident ? 1 : 0)])) return rc;
  const int GX = a.B * a.tilesY * a.tilesX, GY = cdiv(a.Cout, 64);
  a.xsp = GX % 8 == 0 ? 8 : 0; a.xsn = 1;
  a.stat_carry = a.stat_part != nullptr;
  a.stat_rows = (long)grid * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, GX * GY, GX, GY);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

template <int TH, int TW, int BN, int WM, int WN, int NS>
int launch_dma16(Conv3x3Args& a, hipStream_t s) {
  constexpr int ROWS = (TH + 2) * (TW + 4);
  constexpr int HG = ((ROWS + 15) / 16 + 3) / 4;
  constexpr int lds = 2 * HG * 4 * 1024 + NS * BN * 64;      // the epilogue stages inside this (static_assert there)
  static_assert(lds <= 160 * 1024, "LDS budget");
  a.tilesY = cdiv(a.H, TH);
  a.tilesX = cdiv(a.W, TW);
  auto kern = conv3x3_dma16_kernel<TH, TW, BN, WM, WN, NS>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  if (!a.kpart) a.ksplit = 1;
  dim3 grid(a.B * a.tilesY * a.tilesX, cdiv(a.Cout, BN), a.ksplit);
  a.stat_rows = (long)grid.x * (a.direct_ep ? WM : 1);      // (s2s_conv3x3_stat_blocks reports the same count)
  if (a.stat_part && a.stat_rows_req && a.stat_rows_req != a.stat_rows) return S2S_ERR_SHAPE;
  static const int xcd_aware = [] { const char* e = getenv("S2S_CONV_XCD"); return e ? atoi(e) : 1; }();
  a.xsp = 0; a.xsn = 1;
  if (xcd_aware && a.ksplit == 1 && ((long)grid.x * grid.y) % 8 == 0) {
    // bytes the 8 L2s fetch between them: an XCD reads the activations of its pixel range and the weights of its
    // channel range, so the grid cut (sp pixel ranges x sn channel ranges) costs sn * |activations| + sp * |weights|
    const double act = 2.0 * a.B * a.H * a.W * (a.c0 + a.c1), wb = 2.0 * 9 * (a.c0 + a.c1) * a.Cout;
    double best = 0;
    for (int sp = 8; sp >= 1; sp >>= 1) {
      const int sn = 8 / sp;
      if (grid.x % sp || grid.y % sn) continue;
      const double cost = sn * act + sp * wb;
      if (!a.xsp || cost < best) { best = cost; a.xsp = sp; a.xsn = sn; }
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// =========================================================================================================
// KS x KS-tap variants of the loop above with an output size different from the input's (SURVEY.md section 8, row
// a13): y[i][j] = sum_{a,b < KS} W[a][b] x[i + a - PAD][j + b - PAD], input (H + KS - 1 - 2 PAD) squared-ish, a.H / a.W
// the OUTPUT size as everywhere.
//   KS = 2: a 4x4 stride-2 pad-1 convolution is a 2x2 "valid" convolution (PAD = 0, Hin = H + 1) over the
//           space-to-depth image of the padded input (K = 4 taps x 4*Cin, no wasted MACs); its data gradient / the 4x4
//           stride-2 transposed convolution is the same loop with flipped taps and PAD = 1 (Hin = H - 1).
//   KS = 4: the PatchGAN's 4x4 stride-1 pad-1 layers (PAD = 1, Hin = H + 1) and their data gradient (PAD = 2,
//           Hin = H - 1).
// Same halo image, weight ring, swizzle, operand order and epilogue as conv3x3_dma16_kernel; weights packed
// [chunk][tap a*KS+b][Cout][32].  One source tensor.
// =========================================================================================================
// MODE 1 (KS = 2, PAD = 0): the 4x4 stride-2 convolution straight from the PLAIN input [B][2H][2W][C] -- the halo loader
//   does the space-to-depth in its DMA addresses: virtual channel (r*2+s)*C + c of cell (p, q) is channel c of pixel
//   (2p + r - 1, 2q + s - 1), zero outside; a.c0 = 4 C (C a power of two, a.lgc its log2).  No layout pass, no border cells.
// MODE 2 (KS = 2): the 4x4 stride-2 TRANSPOSED convolution (and the stride-2 convolution's data gradient) by sub-pixel
//   phase: phase (r, s) = blockIdx.z / ksplit is a 2x2 convolution with padding (r, s) over the h x w input whose
//   output pixel (i, j) is pixel (2i + 1 - r, 2j + 1 - s) of the plain 2h x 2w output; its weights are rows
//   [phase C, phase C + C) of the packed data-gradient operand (4 C rows).  a.H / a.W = h, w; a.Cout = C.  The four
//   phases tile h x w exactly (the space-to-depth form computes (h+1) x (w+1) cells) and nothing is re-laid out.
template <int TH, int TW, int BN, int WM, int WN, int NS, int KS, int PAD, int MODE = 0, bool SPARSE = false>
__global__ __launch_bounds__(256, 2) void convkxk_dma16_kernel(Conv3x3Args a) {
  static_assert(MODE == 0 || KS == 2, "the fused layouts exist for the 2x2-tap forms");
  using T = bf16_t;
  constexpr int TAPS = KS * KS;
  static_assert(KS >= 2 && KS <= 5, "the halo row has four spare columns");
  constexpr int HP = TW + 4, HH_ = TH + KS - 1, ROWS = HH_ * HP;
  constexpr int NGA = (ROWS + 15) / 16, HG = (NGA + 3) / 4;
  constexpr int A_BYTES = HG * 4 * 1024;
  constexpr int BG = BN / 64;
  constexpr int B_BYTES = BN * 64;
  constexpr int BM = TH * TW, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int RB = TW / 16;
  static_assert(WM * WN == 4 && BN % 64 == 0 && NS == 4 && TW % 16 == 0 && WTM % TW == 0, "configuration");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem;
  char* const ldsB = smem + 2 * A_BYTES;
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const char* __restrict__ wp = static_cast<const char*>(a.w);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int cl = lane & 15, kp = lane >> 4;
  int bt = blockIdx.x;
  const int tx = bt % a.tilesX; bt /= a.tilesX;
  const int ty = bt % a.tilesY;
  const int img = bt / a.tilesY;
  const int y0 = ty * TH, x0p = tx * TW;
  const int n0 = blockIdx.y * BN;
  const int phase = MODE == 2 ? (int)blockIdx.z / a.ksplit : 0, zk = (int)blockIdx.z - phase * a.ksplit;
  const int pad_y = MODE == 2 ? (phase >> 1) : PAD, pad_x = MODE == 2 ? (phase & 1) : PAD;
  const int Hi = MODE == 2 ? a.H : a.H + KS - 1 - 2 * PAD, Wi = MODE == 2 ? a.W : a.W + KS - 1 - 2 * PAD;
  // this workgroup's chunk range [c_lo, c_hi): all of them, or the zk-th share of a split-K launch
  const int c_lo = (int)(((long)zk * a.nchunk) / a.ksplit), c_hi = (int)(((long)(zk + 1) * a.nchunk) / a.ksplit);
  const int wrows = MODE == 2 ? 4 * a.Cout : a.Cout, wrow0 = (MODE == 2 ? phase * a.Cout : 0) + n0;

  const int drow = lane >> 2, dslot = lane & 3;
  int apix[HG], apc[HG];
#pragma unroll
  for (int j = 0; j < HG; ++j) {
    const int row = (wave + 4 * j) * 16 + drow;
    const int hy = row / HP, hx = row - hy * HP;
    const int gy = y0 - pad_y + hy, gx = x0p - pad_x + hx;
    const bool inside = row < ROWS && hx < TW + KS - 1 && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
    if (MODE == 1) apix[j] = inside ? ((gy << 16) | gx) : -1;        // cell coordinates; the pixel depends on the chunk
    else apix[j] = inside ? (img * Hi + gy) * Wi + gx : -1;
    apc[j] = (dslot ^ (((hx >> 2) & 1) << 1)) * 8;      // slot key of the halo column: see conv3x3_dma16_kernel
  }
  const char* wptr[BG];
  int wstep[BG];
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int n = (wave + 4 * j) * 16 + drow;
    const bool ok = n0 + n < a.Cout;
    wptr[j] = ok ? wp + (long)c_lo * TAPS * wrows * 64 + ((long)(wrow0 + n) * 32 + ((dslot ^ ((-(n >> 2)) & 3)) * 8)) * 2 : g_zero_page;
    wstep[j] = ok ? wrows * 64 : 0;
  }
  auto dma_halo = [&](int c) {
    char* dst = ldsA + (c & 1) * A_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      const int ch = c * 32 + apc[j];
      const void* g = g_zero_page;
      if (MODE == 1) {
        if (apix[j] >= 0 && ch < a.c0) {
          const int rs = ch >> a.lgc, cc = ch & ((1 << a.lgc) - 1);
          const int py = 2 * (apix[j] >> 16) + (rs >> 1) - 1, px = 2 * (apix[j] & 0xffff) + (rs & 1) - 1;
          if ((unsigned)py < (unsigned)(2 * a.H) && (unsigned)px < (unsigned)(2 * a.W))
            g = x0 + ((long)(img * 2 * a.H + py) * (2 * a.W) + px) * a.ld0 + cc;
        }
      } else if (apix[j] >= 0 && ch < a.c0) {
        g = x0 + (long)apix[j] * a.ld0 + ch;
      }
      dma16(g, dst + j * 4096);
    }
  };
  auto dma_w = [&](int slot) {
    char* dst = ldsB + slot * B_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      dma16(wptr[j], dst + j * 4096);
      wptr[j] += wstep[j];
    }
  };
  int aofs[RB][KS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      const int px = rb * 16 + cl + kw;
      aofs[rb][kw] = ((wm * (WTM / TW)) * HP + px) * 64 + ((kp ^ (((px >> 2) & 1) << 1)) << 4);
    }
  const int nrow = wn * WTN + cl;
  const int bofs = nrow * 64 + ((kp ^ ((-(nrow >> 2)) & 3)) << 4);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mi][ni][j] = 0.f;

  // SPARSE (launches with Cout <= 32 on a 64-wide tile: the image-side layers, the PatchGAN's single logit): only the
  // 16-channel blocks of this wave's column range that hold real output channels are multiplied -- the MFMAs and fragment
  // reads of the dead blocks were most of those launches' time.  A separate instantiation, so that the dense kernels keep
  // their register allocation (as a run-time branch in the common kernel the 128-wide tiles spilled 880 registers).
  const int nlive = SPARSE ? __builtin_amdgcn_readfirstlane(min(NI, max(0, (a.Cout - n0 - wn * WTN + 15) >> 4))) : NI;
  auto compute = [&](auto tapc, const char* Ab, const char* Bb) {
    constexpr int tap = decltype(tapc)::value;
    constexpr int kh = tap / KS, kw = tap % KS;
    if constexpr (!SPARSE) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const bf16x8*>(Ab + aofs[mi % RB][kw] + (kh + mi / RB) * HP * 64);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(Bb + bofs + ni * 1024);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    } else if (nlive > 0) {
      bf16x8 af[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const bf16x8*>(Ab + aofs[mi % RB][kw] + (kh + mi / RB) * HP * 64);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        if (ni < nlive) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(Bb + bofs + ni * 1024);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, af[mi], acc[mi][ni], 0, 0, 0);
        }
    }
  };
  dma_halo(c_lo);
  static_for<NS - 1>([&](auto k) { dma_w(decltype(k)::value); });     // (c_hi - c_lo) * TAPS >= 3 slabs always exist
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  int c = c_lo;
  for (; c + 1 < c_hi; ++c) {
    const char* Ab = ldsA + (c & 1) * A_BYTES;
    const int it0 = c * TAPS;
    static_for<TAPS>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap == 0) dma_halo(c + 1);
      dma_w((it0 + tap + NS - 1) % NS);
      compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      wait_vm<(NS - 2) * BG + (tap <= NS - 3 ? HG : 0)>();
      __builtin_amdgcn_s_barrier();
    });
  }
  float biasr[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * WTN + ni * 16 + 4 * kp + j;
      biasr[ni][j] = (a.bias && n < a.Cout) ? a.bias[n % a.bias_mod] : 0.f;
    }
  {
    const char* Ab = ldsA + (c & 1) * A_BYTES;
    const int it0 = c * TAPS;
    static_for<TAPS>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap + NS - 1 < TAPS) dma_w((it0 + tap + NS - 1) % NS);
      compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + NS - 1 < TAPS) wait_vm<(NS - 2) * BG>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
    });
  }
  const OutMap om = MODE == 2 ? OutMap{2, 1 - pad_y, 1 - pad_x, 2 * a.H, 2 * a.W} : OutMap{1, 0, 0, a.H, a.W};
  if (a.kpart) {
    // split-K: the fp32 partial tile straight from the accumulators (a lane's four registers = four consecutive
    // channels of one pixel: one 16-byte store)
    float* const kpz = a.kpart + (long)zk * ((long)a.B * om.OH * om.OW) * a.Cout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = wm * WTM + mi * 16 + cl;
      const int py = m / TW, px = m - py * TW;
      const int gy = y0 + py, gx = x0p + px;
      if (gy < a.H && gx < a.W) {
        float* const row = kpz + (((long)img * om.OH + gy * om.os + om.oy) * om.OW + gx * om.os + om.ox) * a.Cout;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int n = n0 + wn * WTN + ni * 16 + 4 * kp;
          if (n < a.Cout) *reinterpret_cast<f32x4*>(row + n) = acc[mi][ni];
        }
      }
    }
    return;
  }
  // (per-tile statistics rows are only defined for the single-phase forms)
  const long stat_row = MODE == 2 ? -1 : (long)blockIdx.x;
  if (!a.stat_part && a.direct_ep) {
    if (a.ep_scale) conv_epilogue16_direct<TH, TW, BN, WM, WN, true>(a, acc, biasr, img, y0, x0p, n0, tid, om);
    else conv_epilogue16_direct<TH, TW, BN, WM, WN, false>(a, acc, biasr, img, y0, x0p, n0, tid, om);
  } else if (a.ep_scale) conv_epilogue16<TH, TW, BN, WM, WN, 2 * A_BYTES + NS * B_BYTES, true>(a, acc, biasr, smem, img, y0, x0p, n0, tid, stat_row, om);
  else conv_epilogue16<TH, TW, BN, WM, WN, 2 * A_BYTES + NS * B_BYTES, false>(a, acc, biasr, smem, img, y0, x0p, n0, tid, stat_row, om);
}

template <int TH, int TW, int BN, int WM, int WN, int KS, int PAD, int MODE = 0, bool SPARSE = false>
int launch_convkxk(Conv3x3Args& a, hipStream_t s) {
  if constexpr (!SPARSE && BN == 64 && MODE == 0) {      // few real output channels on a 64-wide tile: the sparse form
    if (a.Cout <= 32) return launch_convkxk<TH, TW, BN, WM, WN, KS, PAD, MODE, true>(a, s);
  }
  constexpr int NS = 4;
  constexpr int ROWS = (TH + KS - 1) * (TW + 4);
  constexpr int HG = ((ROWS + 15) / 16 + 3) / 4;
  constexpr int lds = 2 * HG * 4 * 1024 + NS * BN * 64;      // the epilogue stages inside this (static_assert there)
  static_assert(lds <= 160 * 1024, "LDS budget");
  a.tilesY = cdiv(a.H, TH);
  a.tilesX = cdiv(a.W, TW);
  auto kern = convkxk_dma16_kernel<TH, TW, BN, WM, WN, NS, KS, PAD, MODE, SPARSE>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  if (!a.kpart) a.ksplit = 1;
  dim3 grid(a.B * a.tilesY * a.tilesX, cdiv(a.Cout, BN), a.ksplit * (MODE == 2 ? 4 : 1));
  a.stat_rows = grid.x;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// =========================================================================================================
// FLAT form of MODE 1 / MODE 2 for the inner U-Net levels (maps of at most 8x8): the output tile is 128 consecutive
// pixels of the FLATTENED batch (pixel p = (n, i, j)), not a 2-D window of one image.  A 4x4 map fills 16 of the 128
// pixels of an 8x16 window, and every image's workgroups then stream the layer's whole 8-17 MB weight operand from L2
// (16 images: 130-270 MB per launch, 20-70 us for 0.1-17 GFLOP); here the batch shares tiles and the operand is
// streamed once per 128 pixels.  The halo image is replaced by an im2col image in LDS -- one [128 pixels][32 ch]
// plane per tap, filled by per-lane DMA addresses -- so the tap is again a constant LDS offset.  Always split-K: the
// fp32 partial tile goes to kpart[z][output pixel][C] and convk_splitk_reduce_kernel finishes (bias, activation).
//   MODE 1: conv 4x4 s2 from the plain input [B][2h][2w][Cin] (a.c0 = 4 Cin virtual channels), a.H / a.W = h, w output
//   MODE 2: transposed conv by phase, a.H / a.W = h, w input, plain output [B][2h][2w][C]
// h, w powers of two (a.lgh, a.lgw).
// =========================================================================================================
// These launches are latency-bound, not MFMA-bound (a tap is ~0.1 us of matrix work, its weight slab an HBM round trip),
// so the weight ring is NS = 8 slots deep -- seven taps of prefetch -- and the loop is uniform: slabs and planes past the
// workgroup's range are fetched from the zero page, which keeps every s_waitcnt count a compile-time constant.
template <int BN, int MODE, int NS>
__global__ __launch_bounds__(256, 1) void convflat_dma16_kernel(Conv3x3Args a) {
  using T = bf16_t;
  constexpr int TAPS = 4, WM = 2, WN = 2, BM = 128;
  constexpr int HG = TAPS * BM / 16 / 4;               // DMA instructions per wave and chunk for the im2col planes (8)
  constexpr int PLANE = BM * 64, A_BYTES = TAPS * PLANE;
  constexpr int BG = BN / 64, B_BYTES = BN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  static_assert(MODE == 1 || MODE == 2, "flat forms of the layout-free kernels");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem;
  char* const ldsB = smem + 2 * A_BYTES;
  const T* __restrict__ x0 = static_cast<const T*>(a.x0);
  const char* __restrict__ wp = static_cast<const char*>(a.w);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int cl = lane & 15, kp = lane >> 4;
  const int p0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int h = 1 << a.lgh, w = 1 << a.lgw;
  const int npix = a.B << (a.lgh + a.lgw);
  const int phase = MODE == 2 ? (int)blockIdx.z / a.ksplit : 0, zk = (int)blockIdx.z - phase * a.ksplit;
  const int pad_y = MODE == 2 ? (phase >> 1) : 0, pad_x = MODE == 2 ? (phase & 1) : 0;
  const int c_lo = (int)(((long)zk * a.nchunk) / a.ksplit), c_hi = (int)(((long)(zk + 1) * a.nchunk) / a.ksplit);
  const int wrows = MODE == 2 ? 4 * a.Cout : a.Cout, wrow0 = (MODE == 2 ? phase * a.Cout : 0) + n0;

  const int drow = lane >> 2, dslot = lane & 3;
  int apix[HG], apc[HG];
#pragma unroll
  for (int j = 0; j < HG; ++j) {
    const int row = (wave + 4 * j) * 16 + drow;               // LDS row: plane = tap, row inside the plane = tile pixel
    const int tap = row >> 7, m = row & (BM - 1);
    const int p = p0 + m;
    const int n = p >> (a.lgh + a.lgw), iy = (p >> a.lgw) & (h - 1), ix = p & (w - 1);
    const int yy = iy + (tap >> 1) - pad_y, xx = ix + (tap & 1) - pad_x;      // MODE 1: cell; MODE 2: input pixel
    if (MODE == 1) apix[j] = p < npix ? ((n << 8) | (yy << 4) | xx) : -1;
    else apix[j] = (p < npix && yy >= 0 && yy < h && xx >= 0 && xx < w) ? (n * h + yy) * w + xx : -1;
    apc[j] = (dslot ^ ((-(m >> 2)) & 3)) * 8;
  }
  const char* wptr[BG];
  int wstep[BG];
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int n = (wave + 4 * j) * 16 + drow;
    const bool ok = n0 + n < a.Cout;
    wptr[j] = ok ? wp + (long)c_lo * TAPS * wrows * 64 + ((long)(wrow0 + n) * 32 + ((dslot ^ ((-(n >> 2)) & 3)) * 8)) * 2 : g_zero_page;
    wstep[j] = ok ? wrows * 64 : 0;
  }
  auto dma_planes = [&](int c) {
    char* dst = ldsA + (c & 1) * A_BYTES + wave * 1024;
    const bool live = c < c_hi;
#pragma unroll
    for (int j = 0; j < HG; ++j) {
      const int ch = c * 32 + apc[j];
      const void* g = g_zero_page;
      if (!live) {
      } else if (MODE == 1) {
        if (apix[j] >= 0 && ch < a.c0) {
          const int rs = ch >> a.lgc, cc = ch & ((1 << a.lgc) - 1);
          const int py = 2 * ((apix[j] >> 4) & 15) + (rs >> 1) - 1, px = 2 * (apix[j] & 15) + (rs & 1) - 1;
          if ((unsigned)py < (unsigned)(2 * h) && (unsigned)px < (unsigned)(2 * w))
            g = x0 + ((long)((apix[j] >> 8) * 2 * h + py) * (2 * w) + px) * a.ld0 + cc;
        }
      } else if (apix[j] >= 0 && ch < a.c0) {
        g = x0 + (long)apix[j] * a.ld0 + ch;
      }
      dma16(g, dst + j * 4096);
    }
  };
  const int total = (c_hi - c_lo) * TAPS;              // weight slabs of this workgroup
  auto dma_w = [&](int slot, int it) {
    char* dst = ldsB + slot * B_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      dma16(it < total ? wptr[j] : g_zero_page, dst + j * 4096);
      wptr[j] += wstep[j];
    }
  };
  int aofs[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 16 + cl;
    aofs[mi] = m * 64 + ((kp ^ ((-(m >> 2)) & 3)) << 4);
  }
  const int nrow = wn * WTN + cl;
  const int bofs = nrow * 64 + ((kp ^ ((-(nrow >> 2)) & 3)) << 4);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mi][ni][j] = 0.f;

  auto compute = [&](auto tapc, const char* Ab, const char* Bb) {
    constexpr int tap = decltype(tapc)::value;
    bf16x8 af[MI], bfr[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(Ab + tap * PLANE + aofs[mi]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(Bb + bofs + ni * 1024);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
  };
  dma_planes(c_lo);
  static_for<NS - 1>([&](auto k) { dma_w(decltype(k)::value, decltype(k)::value); });
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  for (int c = c_lo; c < c_hi; ++c) {
    const char* Ab = ldsA + (c & 1) * A_BYTES;
    const int it0 = (c - c_lo) * TAPS;
    static_for<TAPS>([&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      if (tap == 0) dma_planes(c + 1);
      dma_w((it0 + tap + NS - 1) % NS, it0 + tap + NS - 1);
      compute(tapc, Ab, ldsB + ((it0 + tap) % NS) * B_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      // slab it+1 must have landed: younger than it are the NS-2 slabs of the iterations since, plus the plane batches
      // issued in those iterations (at their tap 0); the last tap of a chunk also needs the next chunk's planes,
      // which only the slabs of this chunk are younger than
      constexpr int planes_in_flight = [] { int n = 0; for (int k = 0; k <= NS - 3; ++k) n += (((tap - k) % TAPS + TAPS) % TAPS) == 0; return n; }();
      constexpr int allow = (NS - 2) * BG + planes_in_flight * HG;
      constexpr int allow_last = TAPS * BG < allow ? TAPS * BG : allow;
      wait_vm<(tap == TAPS - 1) ? allow_last : allow>();
      __builtin_amdgcn_s_barrier();
    });
  }
  wait_vm<0>();
  // fp32 partial tile -> kpart[zk][output pixel][C]
  const long out_pix = MODE == 2 ? 4L * npix : (long)npix;
  float* const kpz = a.kpart + (long)zk * out_pix * a.Cout;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = p0 + wm * WTM + mi * 16 + cl;
    if (p < npix) {
      long op = p;
      if (MODE == 2) {
        const int n = p >> (a.lgh + a.lgw), iy = (p >> a.lgw) & (h - 1), ix = p & (w - 1);
        op = ((long)n * 2 * h + 2 * iy + 1 - pad_y) * (2 * w) + 2 * ix + 1 - pad_x;
      }
      float* const row = kpz + op * a.Cout;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * WTN + ni * 16 + 4 * kp;
        if (n < a.Cout) *reinterpret_cast<f32x4*>(row + n) = acc[mi][ni];
      }
    }
  }
}

template <int BN, int MODE, int NS = 4>       // (an eight-slot ring, 128 instead of 96 KB of LDS, measured no faster in round 3)
int launch_convflat(Conv3x3Args& a, hipStream_t s) {
  constexpr int lds = 2 * 4 * 128 * 64 + NS * BN * 64;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = convflat_dma16_kernel<BN, MODE, NS>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  const int npix = a.B << (a.lgh + a.lgw);
  dim3 grid(cdiv(npix, 128), cdiv(a.Cout, BN), a.ksplit * (MODE == 2 ? 4 : 1));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// out[p][n] = act(sum_z kpart[z][p][n] + bias[n % bias_mod]) for the split-K launches; 8 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void convk_splitk_reduce_kernel(Conv3x3Args a, long npix) {
  const int cp = a.Cout >> 3;
  const long total = npix * cp, slab = npix * a.Cout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / cp;
    const int n = (int)(i - p * cp) * 8;
    f32x8 acc;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc.v[k] = a.bias ? a.bias[(n + k) % a.bias_mod] : 0.f;
    for (int z = 0; z < a.ksplit; ++z) {
      const f32x8 v = load8(a.kpart + (long)z * slab + p * a.Cout + n);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc.v[k] += v.v[k];
    }
    f32x8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (a.ep_scale) {                                  // eval-mode BatchNorm folded behind the 3x3 convolution
        acc.v[k] = fmaf(acc.v[k], a.ep_scale[n + k], a.ep_shift[n + k]);
        if (a.relu) acc.v[k] = fmaxf(acc.v[k], 0.f);
      }
      if (a.act) acc.v[k] = acc.v[k] > 0.f ? acc.v[k] : a.act_slope * acc.v[k];
      r.v[k] = fmaxf(acc.v[k], 0.f);
    }
    store8(static_cast<T*>(a.y) + p * a.ldy + n, acc);
    if (a.y2) store8(static_cast<T*>(a.y2) + p * a.ldy2 + n, r);
  }
}

#ifdef S2S_ABLATE
template <int TH, int TW, int BN, int WM, int WN, int NS>
int launch_dma(Conv3x3Args& a, hipStream_t s) {
  constexpr int ROWS = (TH + 2) * (TW + 4);
  constexpr int HG = ((ROWS + 15) / 16 + 3) / 4;
  constexpr int lds_main = 2 * HG * 4 * 1024 + NS * BN * 64;
  constexpr int RS_ = BN * 2 + 64;
  constexpr int red_ = WM * 2 * BN * 4;
  constexpr int EP_ = (TH * TW * RS_ + red_ <= lds_main) ? 1 : ((TH * TW / 2) * RS_ + red_ <= lds_main ? 2 : 4);
  constexpr int lds_epi = (TH * TW / EP_) * RS_ + red_;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  a.tilesY = cdiv(a.H, TH);
  a.tilesX = cdiv(a.W, TW);
  auto kern = conv3x3_dma_kernel<TH, TW, BN, WM, WN, NS>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(a.B * a.tilesY * a.tilesX, cdiv(a.Cout, BN));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

#endif  // S2S_ABLATE

template <typename T, int TH, int TW, int BN, int WM, int WN, int KS = 3, int PAD = 1>
int launch_cfg(Conv3x3Args& a, hipStream_t s) {
  constexpr bool SPLIT = std::is_same<T, float>::value;
  constexpr int NIMG = SPLIT ? 3 : 1;
  constexpr int lds_main = 2 * NIMG * ((TH + KS - 1) * (TW + KS - 1) + BN) * ROWB;
  constexpr int RS_ = BN * (int)sizeof(T) + (SPLIT ? 16 : 64);
  constexpr int red_ = WM * 2 * BN * 4;
  constexpr int EP_ = (TH * TW * RS_ + red_ <= lds_main) ? 1 : ((TH * TW / 2) * RS_ + red_ <= lds_main ? 2 : 4);
  constexpr int lds_epi = (TH * TW / EP_) * RS_ + red_;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  a.tilesY = cdiv(a.H, TH);
  a.tilesX = cdiv(a.W, TW);
  auto kern = conv3x3_mfma_kernel<T, TH, TW, BN, WM, WN, KS, PAD>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(a.B * a.tilesY * a.tilesX, cdiv(a.Cout, BN));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// =========================================================================================================
// stem: Conv3x3(pad 1) from the NCHW fp32 image (Cin <= 6, three channels per K = 32 step) on the same MFMA path.  The im2col patch of a
// 16x16 pixel tile is built in LDS as a [256][32] image (k = ci*9 + tap), the weights as [BN][32]; one
// K = 32 step per 32x32 block, then the common epilogue (bias, BatchNorm partial sums, coalesced NHWC store).
// The layer is bound by writing its output; T = float uses the three-way bf16 split like the main kernels.
// =========================================================================================================
template <typename T, int BN>
__global__ __launch_bounds__(256) void stem_mfma_kernel(Conv3x3Args a, const float* __restrict__ x,
                                                        const float* __restrict__ w, int Cin) {
  constexpr bool SPLIT = std::is_same<T, float>::value;
  constexpr int NIMG = SPLIT ? 3 : 1;
  constexpr int TH = 16, TW = 16, BM = 256, WM = 4, WN = 1, MI = 2, NI = BN / 32;
  constexpr int P_BYTES = BM * ROWB, W_BYTES = BN * ROWB;
  constexpr int LDS_MAIN = NIMG * (P_BYTES + W_BYTES) + STEM_HALO_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsP = smem;                                   // [NIMG][256][80 B]
  char* const ldsW = smem + NIMG * P_BYTES;                  // [NIMG][BN][80 B]
  float* const xl = reinterpret_cast<float*>(smem + NIMG * (P_BYTES + W_BYTES));   // [Cin][18][18]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int bt = blockIdx.x;
  const int tx = bt % a.tilesX; bt /= a.tilesX;
  const int ty = bt % a.tilesY;
  const int img = bt / a.tilesY;
  const int y0 = ty * TH, x0p = tx * TW, n0 = blockIdx.y * BN;
  const int Kall = Cin * 9;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;
  // input channels in groups of three (27 of the 32 k-columns); RGB is one pass, RGB + mask channel two
  for (int c0 = 0; c0 < Cin; c0 += 3) {
  const int nc = (Cin - c0) < 3 ? (Cin - c0) : 3, K = nc * 9;
  for (int i = tid; i < nc * 324; i += 256) {
    const int ci = i / 324, rr = i - ci * 324, hy = rr / 18, hx = rr - hy * 18;
    const int gy = y0 - 1 + hy, gx = x0p - 1 + hx;
    xl[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                ? x[(((long)img * Cin + c0 + ci) * a.H + gy) * a.W + gx] : 0.f;
  }
  for (int i = tid; i < BN * 4; i += 256) {          // weight rows, 8 k per piece
    const int n = i >> 2, pc = i & 3;
    Piece<float> pw;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kk = pc * 8 + k;
      const float v = (n0 + n < a.Cout && kk < K) ? w[(long)(n0 + n) * Kall + c0 * 9 + kk] : 0.f;
      if (k < 4) pw.a[k] = v; else pw.b[k - 4] = v;
    }
    if constexpr (SPLIT) pw.to_lds(ldsW, W_BYTES, n * ROWB + pc * 16);
    else {
      bf16x8 hv;
#pragma unroll
      for (int k = 0; k < 8; ++k) hv[k] = (bf16_t)(k < 4 ? pw.a[k] : pw.b[k - 4]);
      *reinterpret_cast<bf16x8*>(ldsW + n * ROWB + pc * 16) = hv;
    }
  }
  __syncthreads();
  {                                                    // im2col row of pixel `tid`
    const int py = tid >> 4, px = tid & 15;
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      Piece<float> pp;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = pc * 8 + k;
        float v = 0.f;
        if (kk < K) {
          const int ci = kk / 9, t = kk - ci * 9;
          v = xl[ci * 324 + (py + t / 3) * 18 + px + t % 3];
        }
        if (k < 4) pp.a[k] = v; else pp.b[k - 4] = v;
      }
      if constexpr (SPLIT) pp.to_lds(ldsP, P_BYTES, tid * ROWB + pc * 16);
      else {
        bf16x8 hv;
#pragma unroll
        for (int k = 0; k < 8; ++k) hv[k] = (bf16_t)(k < 4 ? pp.a[k] : pp.b[k - 4]);
        *reinterpret_cast<bf16x8*>(ldsP + tid * ROWB + pc * 16) = hv;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 af[NIMG][MI], bfr[NIMG][NI];
#pragma unroll
    for (int im = 0; im < NIMG; ++im) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[im][mi] = *reinterpret_cast<const bf16x8*>(ldsP + im * P_BYTES + (wave * 64 + mi * 32 + r) * ROWB + ks * 32 + h * 16);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bfr[im][ni] = *reinterpret_cast<const bf16x8*>(ldsW + im * W_BYTES + (ni * 32 + r) * ROWB + ks * 32 + h * 16);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if constexpr (SPLIT) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][mi], bfr[1][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][mi], bfr[0][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mi], bfr[2][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][mi], bfr[0][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mi], bfr[1][ni], acc[mi][ni], 0, 0, 0);
        }
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mi], bfr[0][ni], acc[mi][ni], 0, 0, 0);
      }
  }
  __syncthreads();
  }   // channel groups
  conv_epilogue<T, TH, TW, BN, WM, WN, LDS_MAIN>(a, acc, smem, img, y0, x0p, n0);
}

template <typename T>
int launch_stem(Conv3x3Args& a, const float* x, const float* w, int Cin, hipStream_t s) {
  constexpr bool SPLIT = std::is_same<T, float>::value;
  constexpr int NIMG = SPLIT ? 3 : 1, BN = 64;
  constexpr int lds_main = NIMG * (256 + BN) * ROWB + STEM_HALO_BYTES;
  constexpr int RS_ = BN * (int)sizeof(T) + (SPLIT ? 16 : 64);
  constexpr int red_ = 4 * 2 * BN * 4;
  constexpr int EP_ = (256 * RS_ + red_ <= lds_main) ? 1 : (128 * RS_ + red_ <= lds_main ? 2 : 4);
  constexpr int lds_epi = (256 / EP_) * RS_ + red_;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  a.tilesY = cdiv(a.H, 16);
  a.tilesX = cdiv(a.W, 16);
  auto kern = stem_mfma_kernel<T, BN>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs)) return rc;
  dim3 grid(a.B * a.tilesY * a.tilesX, cdiv(a.Cout, BN));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, x, w, Cin);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// ---- tile configuration table -----------------------------------------------------------------------
// The largest tile whose grid still gives every CU about two workgroups wins; small feature maps
// (16x16 ... 32x32 at batch 16) fall through to smaller tiles so that the 256 CUs stay busy.
struct TileCfg { int th, tw, bn; };
constexpr TileCfg kBf16Cfg[8] = {{8, 32, 128}, {8, 32, 64}, {4, 32, 128}, {4, 32, 64},
                                 {16, 16, 128}, {16, 16, 64}, {8, 16, 128}, {8, 16, 64}};
constexpr TileCfg kF32Cfg[2] = {{4, 32, 64}, {8, 16, 64}};
constexpr int kMinBlocks = 448;

inline long cfg_blocks(const TileCfg& c, int B, int H, int W, int Cout) {
  return (long)B * cdiv(H, c.th) * cdiv(W, c.tw) * cdiv(Cout, c.bn);
}

int select_cfg(int dtype, int B, int H, int W, int Cout) {
  if (dtype == S2S_F32) return W <= 16 ? 1 : 0;
  const char* force = getenv("S2S_CONV_CFG");
  if (force && force[0] >= '0' && force[0] <= '7') return force[0] - '0';
  const int first = W <= 16 ? 4 : 0;
  int best = -1;
  long best_blocks = -1;
  for (int i = first; i < first + 4; ++i) {
    const TileCfg& c = kBf16Cfg[i];
    if (c.bn == 128 && (Cout <= 64 || (Cout % 128 != 0 && Cout % 128 <= 64 && Cout < 512))) continue;   // half-empty N tile
    const long nb = cfg_blocks(c, B, H, W, Cout);
    if (nb >= kMinBlocks) return i;
    if (nb > best_blocks) { best_blocks = nb; best = i; }
  }
  return best;
}

// Does this bf16 launch run on the persistent kernel?  (dispatch() and s2s_conv3x3_stat_rows() share the decision: the
// number of statistics rows follows it.)
inline bool pers_eligible(const Conv3x3Args& a, int id) {
  static const int pers = [] { const char* e = getenv("S2S_CONV_PERS"); return e ? atoi(e) : 1; }();
  // (only where a workgroup gets at least two tiles on average: with one tile each the walk buys nothing and its
  //  bookkeeping costs 3-10 % -- measured on the 64^2 / 32^2 levels, whose grids are ~512 tiles)
  if (id == 0 || id == 4) return false;      // 256-pixel x 128-channel tiles spill in the tile walk: one-tile kernel
  const TileCfg& tc = kBf16Cfg[id];
  const bool many = cfg_blocks(tc, a.B, a.H, a.W, a.Cout) >= (pers > 1 ? pers : 1024);
  return pers && many && a.direct_ep && !a.bias && !a.ep_scale && !a.kpart && !a.act && !a.y2 && a.nchunk >= 2 && !(a.dbg & 64) &&
         a.c0 % 32 == 0 &&                      // a chunk reads one source; 32-bit byte offsets into each source
         (double)a.B * a.H * a.W * a.ld0 * 2 < 4.0e9 && (double)a.B * a.H * a.W * (a.c1 ? a.ld1 : 0) * 2 < 4.0e9;
}

int dispatch(int dtype, Conv3x3Args& a, hipStream_t s) {
  const int id = select_cfg(dtype, a.B, a.H, a.W, a.Cout);
  if (dtype == S2S_F32) {
    // three bf16 images per operand: small tiles so the double-buffered LDS still fits
    if (id == 1) return launch_cfg<float, 8, 16, 64, 2, 2>(a, s);
    return launch_cfg<float, 4, 32, 64, 2, 2>(a, s);
  }
  // bf16: the LDS-DMA loop on v_mfma_f32_16x16x32_bf16.  (Ablation builds, -DS2S_ABLATE, additionally honour
  // S2S_CONV_DMA = 1/3: the 32x32x16 form with a 4/3-slot ring, 0: the register-staged v1 loop.)
#ifdef S2S_ABLATE
  static const int use_dma = [] { const char* e = getenv("S2S_CONV_DMA"); return e ? atoi(e) : 16; }();
  if (use_dma != 16 && use_dma) {
    switch (id) {
      case 0: return use_dma == 3 ? launch_dma<8, 32, 128, 2, 2, 3>(a, s) : launch_dma<8, 32, 128, 2, 2, 4>(a, s);
      case 1: return use_dma == 3 ? launch_dma<8, 32, 64, 4, 1, 3>(a, s) : launch_dma<8, 32, 64, 4, 1, 4>(a, s);
      case 2: return use_dma == 3 ? launch_dma<4, 32, 128, 2, 2, 3>(a, s) : launch_dma<4, 32, 128, 2, 2, 4>(a, s);
      case 3: return use_dma == 3 ? launch_dma<4, 32, 64, 2, 2, 3>(a, s) : launch_dma<4, 32, 64, 2, 2, 4>(a, s);
      case 4: return use_dma == 3 ? launch_dma<16, 16, 128, 2, 2, 3>(a, s) : launch_dma<16, 16, 128, 2, 2, 4>(a, s);
      case 5: return use_dma == 3 ? launch_dma<16, 16, 64, 4, 1, 3>(a, s) : launch_dma<16, 16, 64, 4, 1, 4>(a, s);
      case 6: return use_dma == 3 ? launch_dma<8, 16, 128, 2, 2, 3>(a, s) : launch_dma<8, 16, 128, 2, 2, 4>(a, s);
      case 7: return use_dma == 3 ? launch_dma<8, 16, 64, 2, 2, 3>(a, s) : launch_dma<8, 16, 64, 2, 2, 4>(a, s);
    }
  }
  if (!use_dma) {
    switch (id) {
      case 0: return launch_cfg<bf16_t, 8, 32, 128, 2, 2>(a, s);
      case 1: return launch_cfg<bf16_t, 8, 32, 64, 4, 1>(a, s);
      case 2: return launch_cfg<bf16_t, 4, 32, 128, 2, 2>(a, s);
      case 3: return launch_cfg<bf16_t, 4, 32, 64, 2, 2>(a, s);
      case 4: return launch_cfg<bf16_t, 16, 16, 128, 2, 2>(a, s);
      case 5: return launch_cfg<bf16_t, 16, 16, 64, 4, 1>(a, s);
      case 6: return launch_cfg<bf16_t, 8, 16, 128, 2, 2>(a, s);
      case 7: return launch_cfg<bf16_t, 8, 16, 64, 2, 2>(a, s);
    }
  }
#endif
  // the training step's launches (forward with BatchNorm statistics, data gradients: no bias, no folded affine, no
  // split-K) run on the persistent kernel; S2S_CONV_PERS=0: one tile per workgroup as before
  // many tiles: the staged kernel, one barrier per chunk (statistics launches only when the caller sized stat_part for it)
  if (const int wg = stage_grid(a, a.stat_part != nullptr)) {
    if (!a.stat_part || a.stat_rows_req == (long)wg * 8) return launch_stage(a, wg, s);
  }
  if (pers_eligible(a, id)) {
    switch (id) {      // (the 256-pixel x 128-channel tiles, ids 0 and 4, need 128 accumulator + 48 fragment registers: with the
                       //  tile loop's state they spill, so they stay on the one-tile-per-workgroup kernel)
      case 1: return launch_pers16<8, 32, 64, 4, 1>(a, s);
      case 2: return launch_pers16<4, 32, 128, 2, 2>(a, s);
      case 3: return launch_pers16<4, 32, 64, 2, 2>(a, s);
      case 5: return launch_pers16<16, 16, 64, 4, 1>(a, s);
      case 6: return launch_pers16<8, 16, 128, 2, 2>(a, s);
      case 7: return launch_pers16<8, 16, 64, 2, 2>(a, s);
      default: break;
    }
  }
  // weight ring depth: four slots everywhere (eight gave the 64-channel tiles nothing: 250 vs 242 us on 192 -> 64, round 3)
  switch (id) {
    case 0: return launch_dma16<8, 32, 128, 2, 2, 4>(a, s);
    case 1: return launch_dma16<8, 32, 64, 4, 1, 4>(a, s);
    case 2: return launch_dma16<4, 32, 128, 2, 2, 4>(a, s);
    case 3: return launch_dma16<4, 32, 64, 2, 2, 4>(a, s);
    case 4: return launch_dma16<16, 16, 128, 2, 2, 4>(a, s);
    case 5: return launch_dma16<16, 16, 64, 4, 1, 4>(a, s);
    case 6: return launch_dma16<8, 16, 128, 2, 2, 4>(a, s);
    case 7: return launch_dma16<8, 16, 64, 2, 2, 4>(a, s);
  }
  return S2S_ERR_SHAPE;
}

}  // namespace

// stem forward, called from s2s_stem_conv3x3_fwd (conv_edge.hip); statistics rows = B * ceil(H/16) * ceil(W/16)
int s2s_internal_stem_fwd(int dtype, const float* x_nchw, const float* w_oihw, const float* bias, void* y, int ldy,
                          float* stat_part, int B, int H, int W, int Cin, int Cout, hipStream_t s) {
  Conv3x3Args a{};
  a.x0 = a.x1 = a.w = nullptr; a.bias = bias; a.y = y; a.stat_part = stat_part;
  a.ep_scale = a.ep_shift = nullptr; a.act = 0; a.act_slope = 0.f; a.y2 = nullptr; a.ldy2 = 8; a.bias_mod = Cout;
  a.ld0 = a.c0 = a.ld1 = a.c1 = 0; a.ldy = ldy;
  a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.nchunk = 1; a.relu = 0; a.dbg = 0; a.direct_ep = direct_ep_default();
  a.tilesX = a.tilesY = 0;
  if (dtype == S2S_BF16) return launch_stem<bf16_t>(a, x_nchw, w_oihw, Cin, s);
  if (dtype == S2S_F32) return launch_stem<float>(a, x_nchw, w_oihw, Cin, s);
  return S2S_ERR_DTYPE;
}

// Number of row-blocks of partial statistics the kernel writes for a (B,H,W,Cout) problem
// (= gridDim.x); the caller sizes stat_part as [2][Cout][blocks] floats.
// out: HOST buffer long[n][4] = {shader clock at entry, at exit, wall clock (100 MHz) at entry, at exit} of the
// first n <= 8192 workgroups of the last conv3x3 launch made with S2S_CONV_DBG=64
extern "C" int s2s_debug_conv_clock(long* out, int n) {
  if (!out) return S2S_ERR_NULL;
  if (n <= 0 || n > 8192) return S2S_ERR_SHAPE;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), (size_t)n * 4 * sizeof(long)) == hipSuccess ? S2S_OK
                                                                                                  : S2S_ERR_LAUNCH;
}

extern "C" int s2s_conv3x3_stat_blocks(int dtype, int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return S2S_ERR_SHAPE;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  const int id = select_cfg(dtype, B, H, W, Cout);
  const TileCfg& c = dtype == S2S_F32 ? kF32Cfg[id] : kBf16Cfg[id];
  const int tiles = B * cdiv(H, c.th) * cdiv(W, c.tw);
  if (dtype == S2S_F32 || !direct_ep_default()) return tiles;
#ifdef S2S_ABLATE
  { static const int use_dma = [] { const char* e = getenv("S2S_CONV_DMA"); return e ? atoi(e) : 16; }();
    if (use_dma != 16) return tiles; }
#endif
  // the register epilogue of the bf16 kernel writes one row per (tile, wave row): WM = 4 for the 64-wide 4 x 1 tiles
  static const int wm_of[8] = {2, 4, 2, 2, 2, 4, 2, 2};
  return tiles * wm_of[id];
}

extern "C" int s2s_conv3x3_nhwc_k(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1,
                                  const void* w_packed, const float* bias, void* y, int ldy, float* stat_part,
                                  const float* ep_scale, const float* ep_shift, int relu, float* kwork, int B, int H,
                                  int W, int Cout, void* stream);
extern "C" int s2s_conv3x3_nhwc_s(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1,
                                  const void* w_packed, const float* bias, void* y, int ldy, float* stat_part, int stat_rows,
                                  const float* ep_scale, const float* ep_shift, int relu, float* kwork, int B, int H,
                                  int W, int Cout, void* stream);

// Split count of s2s_conv3x3_nhwc_k for a launch without statistics (bf16; 1 = not split): at small batch the deep
// levels have a few dozen output tiles and K = 9 x 512..1536, so the chunk range is shared by up to 16 workgroups.
// kwork: float[splits][B*H*W][Cout].
extern "C" int s2s_conv3x3_ksplit(int dtype, int B, int H, int W, int Cout, int cin) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || cin <= 0) return S2S_ERR_SHAPE;
  if (dtype != S2S_BF16) return 1;
#ifdef S2S_ABLATE
  static const int use_dma = [] { const char* e = getenv("S2S_CONV_DMA"); return e ? atoi(e) : 16; }();
  if (use_dma != 16) return 1;
#endif
  const TileCfg& c = kBf16Cfg[select_cfg(dtype, B, H, W, Cout)];
  const long base = cfg_blocks(c, B, H, W, Cout);
  const int nchunk = cdiv(cin, 32);
  if (base >= 128 || nchunk < 8) return 1;
  long sp = (512 + base - 1) / base;
  if (sp > nchunk / 4) sp = nchunk / 4;
  if (sp > 16) sp = 16;
  return sp < 2 ? 1 : (int)sp;
}

extern "C" int s2s_conv3x3_nhwc(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1,
                                const void* w_packed, const float* bias, void* y, int ldy, float* stat_part,
                                const float* ep_scale, const float* ep_shift, int relu, int B, int H, int W,
                                int Cout, void* stream) {
  return s2s_conv3x3_nhwc_k(dtype, x0, ld0, c0, x1, ld1, c1, w_packed, bias, y, ldy, stat_part, ep_scale, ep_shift, relu,
                            nullptr, B, H, W, Cout, stream);
}

static void conv3x3_fill_args(Conv3x3Args& a, int ld0, int c0, int ld1, int c1, const float* bias, int B, int H, int W,
                              int Cout) {
  a.bias = bias;
  a.act = 0; a.act_slope = 0.f; a.y2 = nullptr; a.ldy2 = 8; a.bias_mod = Cout; a.kpart = nullptr; a.ksplit = 1; a.lgc = 0; a.lgh = a.lgw = 0;
  a.ld0 = ld0; a.c0 = c0; a.ld1 = ld1; a.c1 = c1;
  a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.nchunk = cdiv(c0 + c1, 32);
  a.tilesX = a.tilesY = 0;
  // S2S_CONV_DBG: 64 = clock probe, 128 = the earlier LDS slot key (both leave the results unchanged); the
  // result-changing timing ablations (bits 1-32) exist only in -DS2S_ABLATE builds
  { static const int dbg = [] { const char* e = getenv("S2S_CONV_DBG"); return (e ? atoi(e) : 0) & S2S_DBG_MASK; }(); a.dbg = dbg; }
  a.direct_ep = direct_ep_default();
}

// Rows of BatchNorm partial sums a TRAINING-step launch (statistics, no folded affine, no split) of these operands
// writes: s2s_conv3x3_stat_blocks() of them, or -- when the launch runs on the persistent kernel with one channel tile --
// one per (workgroup, wave row).  Size stat_part as float[2][Cout][rows] and pass `rows` to s2s_conv3x3_nhwc_s.
extern "C" int s2s_conv3x3_stat_rows(int dtype, int B, int H, int W, int Cout, int c0, int c1, int ld0, int ld1,
                                     int has_bias) {
  const int legacy = s2s_conv3x3_stat_blocks(dtype, B, H, W, Cout);
  if (legacy < 0 || dtype != S2S_BF16 || c0 <= 0 || c1 < 0) return legacy;
#ifdef S2S_ABLATE
  { static const int use_dma = [] { const char* e = getenv("S2S_CONV_DMA"); return e ? atoi(e) : 16; }();
    if (use_dma != 16) return legacy; }
#endif
  Conv3x3Args a{};
  static const float one = 1.f;
  conv3x3_fill_args(a, ld0, c0, ld1, c1, has_bias ? &one : nullptr, B, H, W, Cout);
  if (const int wg = stage_grid(a, true)) return wg * 8;
  const int id = select_cfg(dtype, B, H, W, Cout);
  if (!pers_eligible(a, id)) return legacy;
  const TileCfg& tc = kBf16Cfg[id];
  const int GX = B * cdiv(H, tc.th) * cdiv(W, tc.tw), GY = cdiv(Cout, tc.bn);
  if (GY != 1) return legacy;
  static const int wm_of[8] = {2, 4, 2, 2, 2, 4, 2, 2};
  return pers_plan(a, GX, GY).grid * wm_of[id];
}

extern "C" int s2s_conv3x3_staged(int dtype, int B, int H, int W, int Cout, int c0, int c1, int ld0, int ld1, int ldy,
                                  int stats) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || c0 <= 0 || c1 < 0) return S2S_ERR_SHAPE;
  if (dtype != S2S_BF16) return 0;
  Conv3x3Args a{};
  conv3x3_fill_args(a, ld0, c0, ld1, c1, nullptr, B, H, W, Cout);
  a.ldy = ldy;
  if (!stage_grid(a, stats != 0)) return 0;
  return c0 + c1 <= 64 ? 2 : 1;
}

extern "C" int s2s_conv3x3_nhwc_k(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1,
                                  const void* w_packed, const float* bias, void* y, int ldy, float* stat_part,
                                  const float* ep_scale, const float* ep_shift, int relu, float* kwork, int B, int H,
                                  int W, int Cout, void* stream) {
  return s2s_conv3x3_nhwc_s(dtype, x0, ld0, c0, x1, ld1, c1, w_packed, bias, y, ldy, stat_part, 0, ep_scale, ep_shift, relu,
                            kwork, B, H, W, Cout, stream);
}

// stat_rows: the row count stat_part was sized for (s2s_conv3x3_stat_rows), 0 = s2s_conv3x3_stat_blocks() of them
extern "C" int s2s_conv3x3_nhwc_s(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1,
                                  const void* w_packed, const float* bias, void* y, int ldy, float* stat_part, int stat_rows,
                                  const float* ep_scale, const float* ep_shift, int relu, float* kwork, int B, int H,
                                  int W, int Cout, void* stream) {
  if (!x0 || !w_packed || !y) return S2S_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || c0 <= 0 || c1 < 0) return S2S_ERR_SHAPE;
  if ((c0 % 8) || (c1 % 8) || (ld0 % 8) || (ld1 % 8) || (c1 > 0 && !x1)) return S2S_ERR_SHAPE;
  // the epilogue stores whole 16-byte pieces (8 bf16 / 4 fp32 channels) guarded by `n < Cout` only
  if ((Cout % 8) || (ldy % 8) || ldy < Cout) return S2S_ERR_SHAPE;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return S2S_ERR_NULL;
  const uintptr_t al = 15;
  if (((uintptr_t)x0 & al) || ((uintptr_t)x1 & al) || ((uintptr_t)w_packed & al) || ((uintptr_t)y & al)) return S2S_ERR_ALIGN;
  if (stat_rows < 0) return S2S_ERR_SHAPE;
  Conv3x3Args a{};
  conv3x3_fill_args(a, ld0, c0, ld1, c1, bias, B, H, W, Cout);
  a.x0 = x0; a.x1 = x1; a.w = w_packed; a.y = y; a.stat_part = stat_part; a.stat_rows_req = stat_part ? stat_rows : 0;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ldy = ldy; a.relu = relu;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  a.ksplit = (kwork && !stat_part) ? s2s_conv3x3_ksplit(dtype, B, H, W, Cout, c0 + c1) : 1;
  a.kpart = a.ksplit > 1 ? kwork : nullptr;
  const int rc = dispatch(dtype, a, s);
  if (rc != S2S_OK || !a.kpart) return rc;
  const long npix = (long)B * H * W, pieces = npix * (Cout / 8);
  long nb = (pieces + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(convk_splitk_reduce_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, a, npix);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// ---- KS x KS-tap convolutions of row a13 -------------------------------------------------------------------------
// Output H x W, y[i][j] = sum_{a,b < ks} W[a][b] x[i + a - pad][j + b - pad] (zero outside), input (H + ks - 1 - 2 pad):
//   ks = 2, pad = 0: the 4x4 stride-2 pad-1 convolution, as a 2x2 "valid" convolution over the space-to-depth image of
//                    the padded input (K = 4 taps x 4 Cin, no wasted MACs);
//   ks = 2, pad = 1: its data gradient = the 4x4 stride-2 transposed convolution (flipped taps: see s2s_pack_conv4x4),
//                    output in space-to-depth form with a one-cell border;
//   ks = 4, pad = 1: the PatchGAN's 4x4 stride-1 pad-1 layers;   ks = 4, pad = 2: their data gradient.
// w_packed: [ceil(cin/32)][ks*ks][Cout][32] in the activation dtype.  dtype bf16 runs the LDS-DMA 16x16x32 loop,
// fp32 the register-staged three-way bf16 split (parity mode).  act / act_slope / y2 / bias_mod (0 = Cout): see
// Conv3x3Args.
extern "C" int s2s_convkxk_stat_blocks(int dtype, int B, int H, int W, int Cout, int ks) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (ks != 2 && ks != 4)) return S2S_ERR_SHAPE;
  if (dtype == S2S_F32) return W > 16 ? B * cdiv(H, 4) * cdiv(W, 32) : B * cdiv(H, 8) * cdiv(W, 16);
  if (dtype != S2S_BF16) return S2S_ERR_DTYPE;
  if (ks == 2) return B * cdiv(H, 8) * cdiv(W, W > 16 ? 32 : 16);
  return W > 16 ? B * cdiv(H, 4) * cdiv(W, 32) : B * cdiv(H, 8) * cdiv(W, 16);
}

// Number of K splits s2s_convkxk_nhwc uses when it is handed a workspace (bf16 only): the inner levels of the pix2pix
// generator have 16 ... 64 output tiles, each reducing over K = 16 x 512 ... 1024 -- a few dozen workgroups streaming
// 8 ... 16 MB of weights for tens of microseconds.  Splitting the chunk range brings the launch to ~512 workgroups.
// kwork: float[splits][B*H*W][Cout].
extern "C" int s2s_convkxk_ksplit(int dtype, int B, int H, int W, int Cout, int cin, int ks) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || cin <= 0 || (ks != 2 && ks != 4)) return S2S_ERR_SHAPE;
  if (dtype != S2S_BF16) return 1;
  const bool wide = W > 16, big = Cout > 64;
  const int th = (ks == 4 && wide) ? 4 : 8, tw = wide ? 32 : 16, bn = big ? 128 : 64;
  const long base = (long)B * cdiv(H, th) * cdiv(W, tw) * cdiv(Cout, bn);
  const int nchunk = cdiv(cin, 32);
  // a tile with a single live 16-channel block (the PatchGAN's logit layer: 512 -> 1, padded to 8) runs a quarter of a
  // tile's MFMAs per tap, so its taps are latency-bound and more, shorter workgroups win: aim at 1024 instead of 512
  const long target = Cout <= 16 ? 1024 : 512;
  if (base >= (Cout <= 16 ? 512 : 192) || nchunk < 8) return 1;
  long sp = (target + base - 1) / base;
  if (sp > nchunk / 4) sp = nchunk / 4;               // at least four chunks (16 or 64 taps) per workgroup
  if (sp > 32) sp = 32;
  return sp < 2 ? 1 : (int)sp;
}

extern "C" int s2s_convkxk_nhwc(int dtype, const void* x, int ldx, int cin, const void* w_packed, const float* bias,
                                int bias_mod, void* y, int ldy, void* y2, int ldy2, int act, float act_slope,
                                float* stat_part, float* kwork, int B, int H, int W, int Cout, int ks, int pad,
                                void* stream) {
  if (!x || !w_packed || !y) return S2S_ERR_NULL;
  if (dtype != S2S_BF16 && dtype != S2S_F32) return S2S_ERR_DTYPE;
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || cin <= 0 || (cin % 8) || (ldx % 8) || (Cout % 8) || (ldy % 8) || ldy < Cout) return S2S_ERR_SHAPE;
  if (y2 && ((ldy2 % 8) || ldy2 < Cout)) return S2S_ERR_SHAPE;
  if (bias_mod < 0 || (bias_mod > 0 && Cout % bias_mod)) return S2S_ERR_SHAPE;
  if (!((ks == 2 && (pad == 0 || pad == 1)) || (ks == 4 && (pad == 1 || pad == 2)))) return S2S_ERR_SHAPE;
  if (H + ks - 1 - 2 * pad < 1 || W + ks - 1 - 2 * pad < 1) return S2S_ERR_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)y & 15) || ((uintptr_t)y2 & 15)) return S2S_ERR_ALIGN;
  if ((long)B * (H + 3) * (W + 3) >= (1L << 31)) return S2S_ERR_SHAPE;     // 32-bit pixel index in the halo loader
  Conv3x3Args a{};
  a.x0 = x; a.x1 = nullptr; a.w = w_packed; a.bias = bias; a.y = y; a.stat_part = stat_part;
  a.ep_scale = a.ep_shift = nullptr; a.act = act ? 1 : 0; a.act_slope = act_slope; a.y2 = y2; a.ldy2 = ldy2;
  a.bias_mod = bias_mod > 0 ? bias_mod : Cout;
  a.ksplit = kwork ? s2s_convkxk_ksplit(dtype, B, H, W, Cout, cin, ks) : 1;
  a.kpart = a.ksplit > 1 ? kwork : nullptr;
  if (a.kpart && stat_part) return S2S_ERR_SHAPE;                   // split launches do not produce statistics
  a.ld0 = ldx; a.c0 = cin; a.ld1 = 8; a.c1 = 0; a.ldy = ldy;
  a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.nchunk = cdiv(cin, 32); a.relu = 0; a.dbg = 0; a.direct_ep = direct_ep_default();
  a.tilesX = a.tilesY = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool wide = W > 16, big = Cout > 64;
  const int form = ks * 4 + pad;        // 8: 2x2 valid, 9: 2x2 pad 1, 17: 4x4 pad 1, 18: 4x4 pad 2
  if (dtype == S2S_F32) {
    switch (form) {
      case 8:  return wide ? launch_cfg<float, 4, 32, 64, 2, 2, 2, 0>(a, s) : launch_cfg<float, 8, 16, 64, 2, 2, 2, 0>(a, s);
      case 9:  return wide ? launch_cfg<float, 4, 32, 64, 2, 2, 2, 1>(a, s) : launch_cfg<float, 8, 16, 64, 2, 2, 2, 1>(a, s);
      case 17: return wide ? launch_cfg<float, 4, 32, 64, 2, 2, 4, 1>(a, s) : launch_cfg<float, 8, 16, 64, 2, 2, 4, 1>(a, s);
      default: return wide ? launch_cfg<float, 4, 32, 64, 2, 2, 4, 2>(a, s) : launch_cfg<float, 8, 16, 64, 2, 2, 4, 2>(a, s);
    }
  }
  int rc;
  switch (form) {
    case 8:
      if (wide) rc = big ? launch_convkxk<8, 32, 128, 2, 2, 2, 0>(a, s) : launch_convkxk<8, 32, 64, 4, 1, 2, 0>(a, s);
      else rc = big ? launch_convkxk<8, 16, 128, 2, 2, 2, 0>(a, s) : launch_convkxk<8, 16, 64, 2, 2, 2, 0>(a, s);
      break;
    case 9:
      if (wide) rc = big ? launch_convkxk<8, 32, 128, 2, 2, 2, 1>(a, s) : launch_convkxk<8, 32, 64, 4, 1, 2, 1>(a, s);
      else rc = big ? launch_convkxk<8, 16, 128, 2, 2, 2, 1>(a, s) : launch_convkxk<8, 16, 64, 2, 2, 2, 1>(a, s);
      break;
    case 17:      // 4-row tiles on wide maps: two workgroups per CU with the 7-row halo
      if (wide) rc = big ? launch_convkxk<4, 32, 128, 2, 2, 4, 1>(a, s) : launch_convkxk<4, 32, 64, 2, 2, 4, 1>(a, s);
      else rc = big ? launch_convkxk<8, 16, 128, 2, 2, 4, 1>(a, s) : launch_convkxk<8, 16, 64, 2, 2, 4, 1>(a, s);
      break;
    default:
      if (wide) rc = big ? launch_convkxk<4, 32, 128, 2, 2, 4, 2>(a, s) : launch_convkxk<4, 32, 64, 2, 2, 4, 2>(a, s);
      else rc = big ? launch_convkxk<8, 16, 128, 2, 2, 4, 2>(a, s) : launch_convkxk<8, 16, 64, 2, 2, 4, 2>(a, s);
  }
  if (rc != S2S_OK || !a.kpart) return rc;
  const long npix = (long)B * H * W, pieces = npix * (Cout / 8);
  long nb = (pieces + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(convk_splitk_reduce_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, a, npix);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

// the two entry points of round 1, kept for their callers: s2s_convkxk_nhwc without activation / second output
extern "C" int s2s_conv2x2_stat_blocks(int B, int H, int W, int Cout) {
  return s2s_convkxk_stat_blocks(S2S_BF16, B, H, W, Cout, 2);
}

extern "C" int s2s_conv2x2_nhwc(int dtype, const void* x, int ldx, int cin, const void* w_packed, const float* bias,
                                void* y, int ldy, float* stat_part, int B, int H, int W, int Cout, int pad,
                                void* stream) {
  if (pad != 0 && pad != 1) return S2S_ERR_SHAPE;
  return s2s_convkxk_nhwc(dtype, x, ldx, cin, w_packed, bias, 0, y, ldy, nullptr, 8, 0, 0.f, stat_part, nullptr, B, H, W, Cout,
                          2, pad, stream);
}

extern "C" int s2s_conv4x4s1_nhwc(int dtype, const void* x, int ldx, int cin, const void* w_packed, const float* bias,
                                  void* y, int ldy, float* stat_part, int B, int H, int W, int Cout, int pad,
                                  void* stream) {
  if (pad != 1 && pad != 2) return S2S_ERR_SHAPE;
  return s2s_convkxk_nhwc(dtype, x, ldx, cin, w_packed, bias, 0, y, ldy, nullptr, 8, 0, 0.f, stat_part, nullptr, B, H, W, Cout,
                          4, pad, stream);
}

// ---- the 4x4 stride-2 layers without a layout pass (bf16) -----------------------------------------------------------
// s2s_conv4x4s2_nhwc: nn.Conv2d(k=4, s=2, p=1) from the PLAIN input x [B][2H][2W][Cin] (Cin a power of two >= 8) to
//   y [B][H][W][Cout]; wf = the forward operand of s2s_pack_conv4x4 (stride 2).  act / y2 / kwork as s2s_convkxk_nhwc.
// s2s_convt4x4s2_nhwc: nn.ConvTranspose2d(k=4, s=2, p=1) from x [B][h][w][Cin] to the PLAIN output y [B][2h][2w][C]
//   (C % 64 == 0: narrower layers keep the space-to-depth form), by sub-pixel phase; wd = the data-gradient operand of
//   s2s_pack_conv4x4 for the weight read as [O = Cin][C][4][4].  The same call is the data gradient of
//   s2s_conv4x4s2_nhwc (x = dY, wd of the convolution's own weight, C = its input channels).
// wide maps (W > 16): 0 = 8x32x128, 1 = 8x32x64, 2 = 4x32x128, 3 = 4x32x64; narrow: 4 = 8x16x128, 5 = 8x16x64.
// The largest tile whose grid still fills the chip (two workgroups per CU) wins; `mult` = phases per tile.
static int s2_tile(int B, int H, int W, int C, int mult, long* blocks) {
  static const int th[6] = {8, 8, 4, 4, 8, 8}, tw[6] = {32, 32, 32, 32, 16, 16}, bn[6] = {128, 64, 128, 64, 128, 64};
  const int first = W > 16 ? 0 : 4, last = W > 16 ? 4 : 6;
  int best = first;
  long bb = -1;
  for (int i = first; i < last; ++i) {
    if (bn[i] == 128 && C <= 64) continue;
    const long nb = (long)mult * B * cdiv(H, th[i]) * cdiv(W, tw[i]) * cdiv(C, bn[i]);
    if (nb >= 448) { best = i; bb = nb; break; }
    if (nb > bb) { bb = nb; best = i; }
  }
  if (blocks) *blocks = bb;
  return best;
}

// The flat form applies to power-of-two maps of at most 8x8 with at least two chunks to split; it always runs split-K
// (>= 2), ~256 workgroups.
static bool s2_flat_ok(int H, int W, int nchunk) {
  return H <= 8 && W <= 8 && (H & (H - 1)) == 0 && (W & (W - 1)) == 0 && nchunk >= 2;
}
static int s2_flat_ksplit(int B, int H, int W, int C, int phases, int nchunk) {
  const long base = (long)phases * cdiv(B * H * W, 128) * cdiv(C, C > 64 ? 128 : 64);
  long sp = (256 + base - 1) / base;
  if (sp > nchunk) sp = nchunk;
  if (sp > 32) sp = 32;
  return sp < 2 ? 2 : (int)sp;
}

static int s2_ksplit(long base, int nchunk) {
  // (300, not 192: G.downs.3 -- 256 workgroups of 128 tap steps, one per CU -- runs 35 instead of 40 us split in two)
  if (base >= 300 || nchunk < 8) return 1;
  long sp = (512 + base - 1) / base;
  if (sp > nchunk / 4) sp = nchunk / 4;
  if (sp > 32) sp = 32;
  return sp < 2 ? 1 : (int)sp;
}

extern "C" int s2s_conv4x4s2_ksplit(int B, int H, int W, int Cout, int Cin) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0) return S2S_ERR_SHAPE;
  if (s2_flat_ok(H, W, cdiv(4 * Cin, 32))) return s2_flat_ksplit(B, H, W, Cout, 1, cdiv(4 * Cin, 32));
  long base;
  s2_tile(B, H, W, Cout, 1, &base);
  return s2_ksplit(base, cdiv(4 * Cin, 32));
}

extern "C" int s2s_conv4x4s2_nhwc(int dtype, const void* x, int ldx, int Cin, const void* wf, const float* bias, void* y,
                                  int ldy, void* y2, int ldy2, int act, float act_slope, float* kwork, int B, int H, int W,
                                  int Cout, void* stream) {
  // y == NULL: leave the split's partial slabs in kwork for a consumer that folds them (s2s_instnorm_lrelu_fwd_split /
  // _bwd_split); only where the launch IS split -- S2S_ERR_NULL otherwise
  const bool defer = !y;
  if (defer) { if (!kwork || s2s_conv4x4s2_ksplit(B, H, W, Cout, Cin) < 2 || act || y2) return S2S_ERR_NULL; y = kwork; ldy = Cout; }
  if (!x || !wf || !y) return S2S_ERR_NULL;
  if (dtype != S2S_BF16) return S2S_ERR_DTYPE;
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin < 8 || (Cin & (Cin - 1)) || (ldx % 8) || ldx < Cin || (Cout % 8) || (ldy % 8) || ldy < Cout) return S2S_ERR_SHAPE;
  if (y2 && ((ldy2 % 8) || ldy2 < Cout)) return S2S_ERR_SHAPE;
  if (H >= 32768 || W >= 32768 || (long)B * 4 * H * W >= (1L << 31)) return S2S_ERR_SHAPE;     // 16-bit cell coordinates in the loader
  if (((uintptr_t)x & 15) || ((uintptr_t)wf & 15) || ((uintptr_t)y & 15) || ((uintptr_t)y2 & 15)) return S2S_ERR_ALIGN;
  Conv3x3Args a{};
  a.x0 = x; a.x1 = nullptr; a.w = wf; a.bias = bias; a.y = y; a.stat_part = nullptr;
  a.ep_scale = a.ep_shift = nullptr; a.act = act ? 1 : 0; a.act_slope = act_slope; a.y2 = y2; a.ldy2 = ldy2;
  a.bias_mod = Cout; a.lgc = __builtin_ctz((unsigned)Cin);
  a.ld0 = ldx; a.c0 = 4 * Cin; a.ld1 = 8; a.c1 = 0; a.ldy = ldy;
  a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.nchunk = cdiv(4 * Cin, 32); a.relu = 0; a.dbg = 0; a.direct_ep = direct_ep_default();
  a.tilesX = a.tilesY = 0;
  a.ksplit = kwork ? s2s_conv4x4s2_ksplit(B, H, W, Cout, Cin) : 1;
  a.kpart = a.ksplit > 1 ? kwork : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (a.kpart && s2_flat_ok(H, W, a.nchunk)) {
    a.lgh = __builtin_ctz((unsigned)H); a.lgw = __builtin_ctz((unsigned)W);
    rc = Cout > 64 ? launch_convflat<128, 1>(a, s) : launch_convflat<64, 1>(a, s);
  } else
  switch (s2_tile(B, H, W, Cout, 1, nullptr)) {
    case 0: rc = launch_convkxk<8, 32, 128, 2, 2, 2, 0, 1>(a, s); break;
    case 1: rc = launch_convkxk<8, 32, 64, 4, 1, 2, 0, 1>(a, s); break;
    case 2: rc = launch_convkxk<4, 32, 128, 2, 2, 2, 0, 1>(a, s); break;
    case 3: rc = launch_convkxk<4, 32, 64, 2, 2, 2, 0, 1>(a, s); break;
    case 4: rc = launch_convkxk<8, 16, 128, 2, 2, 2, 0, 1>(a, s); break;
    default: rc = launch_convkxk<8, 16, 64, 2, 2, 2, 0, 1>(a, s);
  }
  if (rc != S2S_OK || !a.kpart || defer) return rc;
  const long npix = (long)B * H * W, pieces = npix * (Cout / 8);
  long nb = (pieces + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(convk_splitk_reduce_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, a, npix);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

extern "C" int s2s_convt4x4s2_ksplit(int B, int h, int w, int C, int Cin) {
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || Cin <= 0) return S2S_ERR_SHAPE;
  if (s2_flat_ok(h, w, cdiv(Cin, 32))) return s2_flat_ksplit(B, h, w, C, 4, cdiv(Cin, 32));
  long base;
  s2_tile(B, h, w, C, 4, &base);
  return s2_ksplit(base, cdiv(Cin, 32));
}

extern "C" int s2s_convt4x4s2_nhwc(int dtype, const void* x, int ldx, int Cin, const void* wd, const float* bias, void* y,
                                   int ldy, float* kwork, int B, int h, int w, int C, void* stream) {
  const bool defer = !y;                                // as in s2s_conv4x4s2_nhwc
  if (defer) { if (!kwork || s2s_convt4x4s2_ksplit(B, h, w, C, Cin) < 2) return S2S_ERR_NULL; y = kwork; ldy = C; }
  if (!x || !wd || !y) return S2S_ERR_NULL;
  if (dtype != S2S_BF16) return S2S_ERR_DTYPE;
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C % 64) || Cin <= 0 || (Cin % 8) || (ldx % 8) || ldx < Cin || (ldy % 8) || ldy < C) return S2S_ERR_SHAPE;
  if ((long)B * 4 * h * w >= (1L << 31)) return S2S_ERR_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)wd & 15) || ((uintptr_t)y & 15)) return S2S_ERR_ALIGN;
  Conv3x3Args a{};
  a.x0 = x; a.x1 = nullptr; a.w = wd; a.bias = bias; a.y = y; a.stat_part = nullptr;
  a.ep_scale = a.ep_shift = nullptr; a.act = 0; a.act_slope = 0.f; a.y2 = nullptr; a.ldy2 = 8;
  a.bias_mod = C; a.lgc = 0;
  a.ld0 = ldx; a.c0 = Cin; a.ld1 = 8; a.c1 = 0; a.ldy = ldy;
  a.B = B; a.H = h; a.W = w; a.Cout = C; a.nchunk = cdiv(Cin, 32); a.relu = 0; a.dbg = 0; a.direct_ep = direct_ep_default();
  a.tilesX = a.tilesY = 0;
  a.ksplit = kwork ? s2s_convt4x4s2_ksplit(B, h, w, C, Cin) : 1;
  a.kpart = a.ksplit > 1 ? kwork : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (a.kpart && s2_flat_ok(h, w, a.nchunk)) {
    a.lgh = __builtin_ctz((unsigned)h); a.lgw = __builtin_ctz((unsigned)w);
    rc = C > 64 ? launch_convflat<128, 2>(a, s) : launch_convflat<64, 2>(a, s);
  } else
  switch (s2_tile(B, h, w, C, 4, nullptr)) {
    case 0: rc = launch_convkxk<8, 32, 128, 2, 2, 2, 1, 2>(a, s); break;
    case 1: rc = launch_convkxk<8, 32, 64, 4, 1, 2, 1, 2>(a, s); break;
    case 2: rc = launch_convkxk<4, 32, 128, 2, 2, 2, 1, 2>(a, s); break;
    case 3: rc = launch_convkxk<4, 32, 64, 2, 2, 2, 1, 2>(a, s); break;
    case 4: rc = launch_convkxk<8, 16, 128, 2, 2, 2, 1, 2>(a, s); break;
    default: rc = launch_convkxk<8, 16, 64, 2, 2, 2, 1, 2>(a, s);
  }
  if (rc != S2S_OK || !a.kpart || defer) return rc;
  const long npix = (long)B * 4 * h * w, pieces = npix * (C / 8);
  long nb = (pieces + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(convk_splitk_reduce_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, a, npix);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
