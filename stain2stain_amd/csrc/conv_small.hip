// The pix2pix generator's INNER levels: 4x4 stride-2 convolution / transposed convolution
// on maps of at most 8x8 output pixels per sample, with the InstanceNorm + activation that follows (forward) or the
// InstanceNorm + activation backward that follows (data gradient) inside the same launch.
//
// Why a kernel of its own.  These layers are weight-streaming problems (8-17 MB of bf16 weights for 0.1-8.6 GFLOP); the
// windowed / flat-pixel kernels of conv3x3_mfma.hip fill the chip through split-K, i.e. every layer was
//     conv (fp32 partial slabs to HBM) -> [reduce launch] -> InstanceNorm launch (re-reads the slabs)
// at 15-30 us per launch for microseconds of data movement.  Here the tile is SAMPLE-COMPLETE instead:
//     one workgroup = SG whole samples (16 * NB pixel rows) x 16 output channels, all of K,
// so that (i) there is no split-K across workgroups, no slabs and no reduce, and (ii) every pixel of a (sample, channel)
// pair lives in one workgroup: InstanceNorm's statistics, the normalisation, the activation and both outputs (LeakyReLU'd
// for the next encoder level, ReLU'd into the decoder's concatenation buffer) are the epilogue.  The backward form ends in
// the InstanceNorm + activation backward of the layer that PRODUCED the convolution's input (the data gradient's output
// tensor), which needs the same per-(sample, channel) sums.
//
// Loop.  K is cut over the 8 waves of the workgroup by TAP (mode 1: wave w takes taps w, w + 8; mode 2: wave w takes
// sub-pixel phase w & 3 and two of its four taps), so nothing is shared between waves until the end and the main loop
// has no barrier and no LDS: per tap the im2col addresses are formed once and the chunk loop only advances pointers; a
// wave streams its weight fragments (16 output channels x 32 k = 1 KiB contiguous in the packed operand) and its pixel
// fragments (16 pixel rows x 64 B) straight into registers through a 4-deep ring of plain global loads (a ring slot =
// two chunks = one 128-byte line per pixel row), one v_mfma_f32_16x16x32_bf16 per (weight fragment, pixel block).  Operand order as everywhere
// (weights as A, pixels as B): a lane's four accumulator registers are four consecutive channels of one pixel.  The
// eight partial tiles meet in LDS once.
//
// Dead taps.  A 1x1 output map touches 4 of the 16 taps (the others only ever see padding), a transposed convolution of
// a 1x1 map one tap per sub-pixel phase: those weight slabs are never fetched (ky0 / nky / kx0 / nkx below).
//
//   MODE 1: nn.Conv2d(4, stride 2, pad 1) from the plain input [B][2h][2w][Cin], forward operand wf[4 Cin / 32][4][Cout][32]
//           (virtual channel (r*2+s)*Cin + c of cell (p, q) = channel c of pixel (2p + r - 1, 2q + s - 1); tap (a, b) of the
//           2x2 cell window: kernel row 2a + r, column 2b + s).  h x w = OUTPUT map.
//   MODE 2: nn.ConvTranspose2d(4, stride 2, pad 1) (= the stride-2 convolution's data gradient) by sub-pixel phase from
//           [B][h][w][Cin], data-gradient operand wd[Cin / 32][4][4 Cout][32] (rows phase * Cout + n): phase (r, s) reads input
//           pixel (i + a - r, j + b - s) for tap (a, b) and writes output pixel (2i + 1 - r, 2j + 1 - s).  h x w = INPUT map.
// The same two loops serve the backward pass with the operands exchanged (MODE 2 on wd = data gradient of a stride-2
// convolution, MODE 1 on wf = data gradient of a transposed convolution), exactly as s2s_convt4x4s2_nhwc / s2s_conv4x4s2_nhwc.
#include "common.h"
#include <utility>

namespace {

__device__ __attribute__((aligned(256))) unsigned char g_small_zero[256];

template <int... Is, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(std::make_integer_sequence<int, N>{}, f);
}

struct SmallArgs {
  const bf16_t* x; int ldx; int Cin;
  const char* w;              // MODE 1: wf; MODE 2: wd (see above)
  const float* bias;          // [Cout] or null (epi 0 / 1)
  int B, lgh, lgw;            // MODE 1: h x w = output map; MODE 2: h x w = input map
  int Cout;                   // output channels of the launch (gridDim.y * 16)
  int SG;                     // samples per workgroup (SG * rows per sample <= 16 * NB)
  int nchunk;                 // Cin / 32
  int SP;                     // LDSX: LDS rows from one sample to the next
  int nx, ny;                 // tiles: sample groups x 16-channel tiles (the grid is their product, see the kernel)
  int lgntx;
  int t0y, nty, t0x, ntx;     // live taps: MODE 1 kernel rows t0y .. t0y + nty - 1 (of 4), columns likewise;
                              // MODE 2: nty / ntx = live taps per phase and axis (1 on a 1-wide axis, else 2)
  int epi;                    // 0: bias (+ activation); 1: InstanceNorm + activation forward; 2: InstanceNorm + activation backward
  int act;                    // epi 0: LeakyReLU(slope) on the bias-added output
  float slope, eps;
  bf16_t* raw; int ldraw;     // epi 1: the bias-added convolution output as stored (what the backward reads), optional
  bf16_t* y; int ldy;         // epi 0 / 1: output; epi 2: dz of the normed layer, channels [bwd_c0, Cout) -> [0, Cout - bwd_c0)
  bf16_t* y2; int ldy2;       // epi 0 / 1: optional relu copy; epi 2: the plain data gradient of channels [0, bwd_c0)
  float* stats;               // epi 1: out [4][B][Cout] = mean, invstd, invstd, -mean * invstd
  const float* stats_in;      // epi 2: the normed layer's statistics [4][B][Cout - bwd_c0]
  const bf16_t* z; int ldz;   // epi 2: that layer's stored convolution output [B][out pixels][Cout - bwd_c0]
  const bf16_t* g2; int ldg2; // epi 2: optional second gradient (wrt the relu copy), same shape
  int bwd_c0;                 // epi 2: first channel that belongs to the normed tensor (a multiple of 16)
};

typedef __attribute__((address_space(3))) void sm_lds_void_t;
typedef __attribute__((address_space(1))) const void sm_gbl_void_t;

// One (sample group, 16 output channels) tile.  NB = 16-row pixel blocks per accumulator group; MODE 2 has four groups
// (the sub-pixel phases), MODE 1 one.
//
// LDSX: the input of the workgroup's samples is staged ONCE in LDS (LDS-DMA, whole pixel rows of Cin x 2 bytes, one
// barrier) and the pixel fragments are ds_read_b128s of it; only the weights stream from L2.  Without it every tap
// re-reads its im2col'd pixels through L1 / L2 -- 2-4 x the weight bytes on the 4x4 / 8x8 maps, and a CU takes in no more
// than ~70 GB/s whatever the loop does (measured: d5 forward 20.6 us, u2 36 us, both at that rate).  LDS image: row
// rho(pixel) x pitch, pitch = Cin * 2 + 16 (the pad turns consecutive rows into consecutive 16-byte bank slots); rho
// orders a sample's pixels so that the sixteen rows of a fragment are (nearly) consecutive for every tap -- mode 1
// (stride 2: a tap reads pixels of ONE parity class) parity-major, (iy & 1, ix & 1) then (iy >> 1, ix >> 1); mode 2
// row-major -- and samples are SP rows apart with SP = Pin + P when a block holds several samples (P < 16), so that
// their rows do not meet in a bank either.  Padding pixels read a row of zeros.
template <int MODE, int NB, bool LDSX>
__global__ __launch_bounds__(512, 2) void convsm_kernel(SmallArgs a) {
  constexpr int NW = 8, NG = MODE == 2 ? 4 : 1, D = 4, NQ = NG * NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // epilogue arrays (alias the staged input, behind a barrier)
  float* const part = reinterpret_cast<float*>(smem);                  // [wave][block][pixel][channel]
  float* const val = part + NW * NB * 256;                             // reduced tile [q][pixel][channel]
  float* const xh = val + NQ * 256;                                    // epi 2: normalised activations
  float (*const fin)[16][16] = reinterpret_cast<float (*)[16][16]>(xh + NQ * 256);   // per (local sample, channel)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cl = lane & 15, kp = lane >> 4;
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2.  With
  // the plain (sample group, channel tile) order every XCD streamed the layer's WHOLE weight operand (8-17 MB through a
  // 4 MB L2, eight times over from the Infinity Cache); here an XCD owns ny / 8 channel tiles for all sample groups, so
  // it streams an eighth of the weights and re-reads them from its own L2.
  int bx, by;
  {
    const int L = blockIdx.x;
    if ((a.ny & 7) == 0) {
      const int per = a.ny >> 3, j = L >> 3;
      by = (L & 7) * per + j % per;
      bx = j / per;
    } else {
      bx = L % a.nx; by = L / a.nx;
    }
  }
  const int n0 = by * 16;
  const int s0 = bx * a.SG;
  const int h = 1 << a.lgh, w = 1 << a.lgw, lgP = a.lgh + a.lgw, P = 1 << lgP;
  const int Hin = MODE == 1 ? 2 * h : h, Win = MODE == 1 ? 2 * w : w, Pin = Hin * Win;
  const int pitch = a.Cin * 2 + 16, zrow = a.SG * a.SP;

  // this lane's pixel rows: R = mb * 16 + cl -> local sample R >> lgP, pixel R & (P - 1)
  int rn[NB], rsl[NB], ry[NB], rx[NB];
  bool rok[NB];
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) {
    const int R = mb * 16 + cl, sl = R >> lgP, p = R & (P - 1);
    rsl[mb] = sl;
    rn[mb] = s0 + sl;
    rok[mb] = sl < a.SG && rn[mb] < a.B;
    ry[mb] = p >> a.lgw;
    rx[mb] = p & (w - 1);
  }
  if (LDSX) {
    // stage the samples' input: one LDS-DMA instruction = 1 KiB = (a piece of) one pixel row, wave-uniform destination
    const int NI = (a.Cin * 2) >> 10;
    const int nitems = a.SG * Pin * NI;
    for (int k = wave; k < nitems; k += NW) {
      const int r = k / NI, piece = k - r * NI;
      const int sl = r / Pin, pin = r - sl * Pin;
      const int n = s0 + sl;
      if (n >= a.B) continue;                                  // (wave-uniform)
      const int iy = pin / Win, ix = pin - iy * Win;
      const int rho = MODE == 1 ? sl * a.SP + ((iy & 1) * 2 + (ix & 1)) * (Pin >> 2) + (iy >> 1) * (Win >> 1) + (ix >> 1)
                                : sl * a.SP + pin;
      const bf16_t* src = a.x + ((long)(n * Hin + iy) * Win + ix) * a.ldx + piece * 512 + lane * 8;
      __builtin_amdgcn_global_load_lds((sm_gbl_void_t*)src, (sm_lds_void_t*)(smem + (long)rho * pitch + piece * 1024), 16, 0, 0);
    }
    for (int i = tid; i < (pitch >> 4); i += 512) {
      f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(smem + (long)zrow * pitch + i * 16) = z4;
    }
  }
  // K is cut over the waves by TAP: MODE 1 wave w takes taps w, w + 8, ...; MODE 2 wave w takes phase w & 3 and that
  // phase's taps (w >> 2), (w >> 2) + 2.  Inside a tap the chunk loop only advances pointers.
  const int g = MODE == 2 ? (wave & 3) : 0;
  const int ntaps = a.nty * a.ntx;
  const int tfirst = MODE == 2 ? (wave >> 2) : wave, tstep = MODE == 2 ? 2 : NW;
  // A step is a PAIR of 32-channel chunks = 128 B of every pixel row, and the k index of the two MFMAs is permuted so
  // that lane group kp owns 32 contiguous bytes of it (channels 64 p + 16 kp + [0, 8) for the first MFMA, + [8, 16) for
  // the second): the four lane groups of a row read one whole 128-byte line with two back-to-back 16-byte loads.  The
  // weight fragments follow the same permutation: lane group kp reads row cl of slab 2 p + (kp >> 1), bytes
  // 32 (kp & 1) + [0, 16) and + [16, 32).
  const long wchunk = MODE == 2 ? 16L * a.Cout * 64 : 4L * a.Cout * 64;     // bytes from chunk c to chunk c + 1
  const long wlane = (long)(kp >> 1) * wchunk + (long)cl * 64 + (kp & 1) * 32;
  const char* const wbase = a.w + (long)n0 * 64 + wlane;
  const bf16_t* const xlane = a.x + kp * 16;
  const long wstep = 2 * wchunk;
  const int npair = a.nchunk >> 1;

  f32x4 acc[NB];
#pragma unroll
  for (int mb = 0; mb < NB; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mb][j] = 0.f;

  // tap t of this wave -> (ty, tx) of the kernel window
  auto tap_of = [&](int t, int& ty, int& tx) {
    const int q = t >> a.lgntx, rem = t & (a.ntx - 1);
    if (MODE == 1) { ty = a.t0y + q; tx = a.t0x + rem; }
    else { ty = h == 1 ? (g >> 1) : q; tx = w == 1 ? (g & 1) : rem; }
  };
  // ---- weight stream: its own (tap, pair) state ----
  int wt = tfirst, wc = 0;
  const char* wp;
  long ws;
  auto open_w = [&]() {
    int ty, tx;
    tap_of(wt, ty, tx);
    wp = reinterpret_cast<const char*>(g_small_zero);
    ws = 0;
    if (wt < ntaps) {
      if (MODE == 1) wp = wbase + ((long)((ty & 1) * 2 + (tx & 1)) * a.nchunk * 4 + (ty >> 1) * 2 + (tx >> 1)) * a.Cout * 64;
      else wp = wbase + ((long)(ty * 2 + tx) * 4 + g) * a.Cout * 64;
      ws = wstep;
    }
  };
  // ---- pixel stream: global pointers, or byte offsets into the staged LDS image ----
  int xt = tfirst, xc = 0;
  const bf16_t* xp[NB];
  int xs[NB], xo[NB];
  auto open_x = [&]() {
    int ty, tx;
    tap_of(xt, ty, tx);
    const bool live = xt < ntaps;
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
      int iy, ix;
      if (MODE == 1) { iy = 2 * ry[mb] + ty - 1; ix = 2 * rx[mb] + tx - 1; }
      else { iy = ry[mb] + ty - (g >> 1); ix = rx[mb] + tx - (g & 1); }
      const bool ok = live && rok[mb] && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
      if (LDSX) {
        const int rho = MODE == 1 ? rsl[mb] * a.SP + ((iy & 1) * 2 + (ix & 1)) * (Pin >> 2) + (iy >> 1) * (Win >> 1) + (ix >> 1)
                                  : rsl[mb] * a.SP + iy * Win + ix;
        xo[mb] = (ok ? rho : zrow) * pitch + kp * 32;
      } else {
        xp[mb] = reinterpret_cast<const bf16_t*>(g_small_zero);
        xs[mb] = 0;
        if (ok) { xp[mb] = xlane + ((long)(rn[mb] * Hin + iy) * Win + ix) * a.ldx; xs[mb] = 64; }
      }
    }
  };
  open_w();
  open_x();

  bf16x8 Wf[D][2], Xf[D][NB][2];
  auto issue_w = [&](auto slotc) {
    constexpr int slot = decltype(slotc)::value;
    Wf[slot][0] = *reinterpret_cast<const bf16x8*>(wp);
    Wf[slot][1] = *reinterpret_cast<const bf16x8*>(wp + 16);
    wp += ws;
    if (++wc == npair) { wc = 0; wt += tstep; open_w(); }          // (wave-uniform) next tap of this wave
  };
  auto issue_x = [&](auto slotc) {
    constexpr int slot = decltype(slotc)::value;
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
      if (LDSX) {
        const char* q = smem + xo[mb] + xc * 128;
        Xf[slot][mb][0] = *reinterpret_cast<const bf16x8*>(q);
        Xf[slot][mb][1] = *reinterpret_cast<const bf16x8*>(q + 16);
      } else {
        Xf[slot][mb][0] = *reinterpret_cast<const bf16x8*>(xp[mb]);
        Xf[slot][mb][1] = *reinterpret_cast<const bf16x8*>(xp[mb] + 8);
        xp[mb] += xs[mb];
      }
    }
    if (++xc == npair) { xc = 0; xt += tstep; open_x(); }
  };

  int mytaps = 0;
  for (int t = tfirst; t < ntaps; t += tstep) ++mytaps;
  const int nsteps = mytaps * npair;
  sfor<D>([&](auto d) { issue_w(d); });                 // the weight ring fills while the input image lands
  if (LDSX) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  sfor<D>([&](auto d) { issue_x(d); });
  for (int base = 0; base < nsteps; base += D) {
    sfor<D>([&](auto d) {
      constexpr int slot = decltype(d)::value;
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[slot][0], Xf[slot][mb][0], acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[slot][1], Xf[slot][mb][1], acc[mb], 0, 0, 0);
      }
      issue_w(d);
      issue_x(d);
    });
  }
  if (LDSX) __syncthreads();                            // every wave is done with the staged image: the epilogue re-uses it

  // ---- the partial tiles meet in LDS: [wave][block][pixel][channel] ----
#pragma unroll
  for (int mb = 0; mb < NB; ++mb)
    *reinterpret_cast<f32x4*>(&part[((wave * NB + mb) * 16 + cl) * 16 + kp * 4]) = acc[mb];
  __syncthreads();
  const bool normed = a.epi == 2 && n0 >= a.bwd_c0;
  for (int e = tid; e < NQ * 256; e += 512) {
    float s = 0.f;
    if (MODE == 2) {                                             // group q / NB came from waves q / NB and q / NB + 4
      const int gq = e / (NB * 256), r = e - gq * (NB * 256);
      s = part[gq * NB * 256 + r] + part[(gq + 4) * NB * 256 + r];
    } else {
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) s += part[wv * NB * 256 + e];
    }
    if (a.epi != 2 && a.bias) s += a.bias[n0 + (e & 15)];
    if (a.epi == 1) s = (float)(bf16_t)s;                        // the statistics are those of the values as stored
    val[e] = s;
  }
  __syncthreads();

  // row (q, px) -> local sample, output pixel inside the sample; HWo = output pixels per sample
  const int HWo = MODE == 2 ? 4 * P : P;
  auto row_of = [&](int q, int px, int& sl, int& opix) {
    const int g = q / NB, mb = q - g * NB;
    const int R = mb * 16 + px, p = R & (P - 1);
    sl = R >> lgP;
    if (MODE == 2) {
      const int i = p >> a.lgw, j = p & (w - 1);
      opix = (2 * i + 1 - (g >> 1)) * (2 * w) + 2 * j + 1 - (g & 1);
    } else {
      opix = p;
    }
  };

  if (a.epi == 2 && normed) {
    // dzn = act'(zn) (g [+ g2]);  xh = (z - mean) * invstd, both back into LDS
    const int C2 = a.Cout - a.bwd_c0, cn = n0 - a.bwd_c0;
    for (int it = tid; it < NQ * 32; it += 512) {
      const int q = it >> 5, px = (it >> 1) & 15, half = it & 1;
      int sl, opix;
      row_of(q, px, sl, opix);
      const int n = s0 + sl;
      const bool ok = sl < a.SG && n < a.B;
      float* const v = &val[(q * 16 + px) * 16 + half * 8];
      float* const xv = &xh[(q * 16 + px) * 16 + half * 8];
      if (ok) {
        const long pix = (long)n * HWo + opix;
        const f32x8 zv = load8(a.z + pix * a.ldz + cn + half * 8);
        f32x8 hv;
        if (a.g2) hv = load8(a.g2 + pix * a.ldg2 + cn + half * 8);
        const long so = (long)n * C2 + cn + half * 8, BC = (long)a.B * C2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float mu = a.stats_in[so + k], is = a.stats_in[BC + so + k];
          const float zn = fmaf(zv.v[k], is, -mu * is);
          const float gw = a.g2 ? hv.v[k] : 0.f;
          v[k] = zn > 0.f ? v[k] + gw : a.slope * v[k];
          xv[k] = (zv.v[k] - mu) * is;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = 0.f; xv[k] = 0.f; }
      }
    }
    __syncthreads();
  }

  if (a.epi == 1 || normed) {
    // per (local sample, channel) sums over the sample's pixels
    if (tid < 256) {
      const int sl = tid >> 4, co = tid & 15;
      if (sl < a.SG) {
        double s1 = 0.0, s2 = 0.0;
        float shift = 0.f;
        bool first = true;
        for (int g = 0; g < NG; ++g)
          for (int p = 0; p < P; ++p) {
            const int R = sl * P + p;
            const int e = ((g * NB + (R >> 4)) * 16 + (R & 15)) * 16 + co;
            if (a.epi == 1) {
              if (first) { shift = val[e]; first = false; }
              const double d = (double)(val[e] - shift);
              s1 += d; s2 += d * d;
            } else {
              s1 += (double)val[e];
              s2 += (double)val[e] * (double)xh[e];
            }
          }
        if (a.epi == 1) {
          const double ms = s1 / (double)HWo;
          double var = s2 / (double)HWo - ms * ms;
          if (var < 0.0) var = 0.0;
          const double mean = ms + (double)shift;
          const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
          fin[0][sl][co] = (float)mean;
          fin[1][sl][co] = invstd;
          const int n = s0 + sl;
          if (n < a.B) {
            const long i = (long)n * a.Cout + n0 + co, BC = (long)a.B * a.Cout;
            a.stats[i] = (float)mean; a.stats[BC + i] = invstd; a.stats[2 * BC + i] = invstd;
            a.stats[3 * BC + i] = -(float)mean * invstd;
          }
        } else {
          fin[0][sl][co] = (float)s1 / (float)HWo;
          fin[1][sl][co] = (float)s2 / (float)HWo;
        }
      }
    }
    __syncthreads();
  }

  // ---- apply + store: one work item = 8 channels of one pixel row ----
  for (int it = tid; it < NQ * 32; it += 512) {
    const int q = it >> 5, px = (it >> 1) & 15, half = it & 1;
    int sl, opix;
    row_of(q, px, sl, opix);
    const int n = s0 + sl;
    if (sl >= a.SG || n >= a.B) continue;
    const long pix = (long)n * HWo + opix;
    const float* const v = &val[(q * 16 + px) * 16 + half * 8];
    f32x8 o8, r8;
    if (a.epi == 1) {
      f32x8 raw8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float mean = fin[0][sl][half * 8 + k], is = fin[1][sl][half * 8 + k];
        const float zn = fmaf(v[k], is, -mean * is);
        raw8.v[k] = v[k];
        o8.v[k] = zn > 0.f ? zn : a.slope * zn;
        r8.v[k] = fmaxf(zn, 0.f);
      }
      if (a.raw) store8(a.raw + pix * a.ldraw + n0 + half * 8, raw8);
      store8(a.y + pix * a.ldy + n0 + half * 8, o8);
      if (a.y2) store8(a.y2 + pix * a.ldy2 + n0 + half * 8, r8);
    } else if (a.epi == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float t = v[k];
        o8.v[k] = a.act ? (t > 0.f ? t : a.slope * t) : t;
        r8.v[k] = fmaxf(o8.v[k], 0.f);
      }
      store8(a.y + pix * a.ldy + n0 + half * 8, o8);
      if (a.y2) store8(a.y2 + pix * a.ldy2 + n0 + half * 8, r8);
    } else if (normed) {
      const int C2 = a.Cout - a.bwd_c0, cn = n0 - a.bwd_c0;
      const float* const xv = &xh[(q * 16 + px) * 16 + half * 8];
      const long so = (long)n * C2 + cn + half * 8, BC = (long)a.B * C2;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float is = a.stats_in[BC + so + k];
        o8.v[k] = is * (v[k] - fin[0][sl][half * 8 + k] - xv[k] * fin[1][sl][half * 8 + k]);
      }
      store8(a.y + pix * a.ldy + cn + half * 8, o8);
    } else {                                                     // epi 2, plain part: the data gradient itself
#pragma unroll
      for (int k = 0; k < 8; ++k) o8.v[k] = v[k];
      store8(a.y2 + pix * a.ldy2 + n0 + half * 8, o8);
    }
  }
}

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// tile plan of a launch: rows per sample inside one accumulator group, samples per workgroup, pixel blocks, and whether
// the samples' input is staged in LDS (whole pixel rows of a multiple of 1 KiB, image <= 144 KiB) with its sample pitch
struct SmallPlan { int P, SG, NB, ldsx, SP, lds_bytes; };
inline int small_epi_bytes(int mode, int NB) { const int NQ = (mode == 2 ? 4 : 1) * NB; return (8 * NB + 2 * NQ) * 1024 + 2048; }
inline bool small_plan(int mode, int B, int h, int w, int Cin, int Cout, SmallPlan* pl) {
  if (mode != 1 && mode != 2) return false;
  if (B <= 0 || !pow2(h) || !pow2(w) || Cin <= 0 || Cout <= 0 || (Cin % 64) || (Cout % 16)) return false;
  const int P = h * w;
  if (mode == 1) { if (P > 64) return false; }
  else if (P > 16) return false;                 // MODE 2: 4 P output pixels <= 64 per sample
  const int Pin = mode == 1 ? 4 * P : P;
  // staged form first: SG samples whose input fits LDS
  if ((Cin * 2) % 1024 == 0) {
    int SG, NB;
    if (P >= 16) { SG = B >= 2 ? 2 : 1; NB = SG * P / 16; }
    else { SG = 16 / P; NB = 1; }
    const int SP = (mode == 1 && P < 16) ? Pin + P : Pin;        // (mode 2: a block's samples are P rows apart already)
    const long xbytes = ((long)SG * SP + 1) * (Cin * 2 + 16);
    if (NB <= 2 && xbytes <= 144 * 1024) {
      const int epi = small_epi_bytes(mode, NB);
      pl->P = P; pl->SG = SG; pl->NB = NB; pl->ldsx = 1; pl->SP = SP;
      pl->lds_bytes = (int)(xbytes > epi ? xbytes : epi);
      return true;
    }
  }
  if (mode == 2 && P > 16) return false;
  int SG, NB;
  if (P >= 16) { NB = P / 16; SG = 1; if (mode == 1 && P == 16 && B >= 2) { NB = 2; SG = 2; } }
  else { NB = 1; SG = 16 / P; }
  if (NB == 3 || (mode == 2 && NB != 1)) return false;
  pl->P = P; pl->SG = SG; pl->NB = NB; pl->ldsx = 0; pl->SP = 0; pl->lds_bytes = small_epi_bytes(mode, NB);
  return true;
}

template <int MODE, int NB, bool LDSX>
int launch_small(SmallArgs& a, int lds, hipStream_t s) {
  auto kern = convsm_kernel<MODE, NB, LDSX>;
  static unsigned long long attr_devs = 0;   // hipFuncSetAttribute is per device
  if (int rc = s2s_allow_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, &attr_devs)) return rc;
  hipLaunchKernelGGL(kern, dim3(a.nx * a.ny), dim3(512), lds, s, a);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}

}  // namespace

// 1 when (mode, shape) is one of the sample-complete launches below: h x w a power-of-two map (mode 1: the OUTPUT map of
// the stride-2 convolution, at most 64 pixels; mode 2: the INPUT map of the transposed form, at most 16 pixels),
// Cin % 64 == 0, Cout % 16 == 0, bf16.  Returns 0 (not taken), 1 (taken, pixels streamed from L2) or 2 (taken, input staged in LDS).
extern "C" int s2s_convsm_ok(int dtype, int mode, int B, int h, int w, int Cin, int Cout) {
  SmallPlan pl;
  if (dtype != S2S_BF16 || !small_plan(mode, B, h, w, Cin, Cout, &pl)) return 0;
  return pl.ldsx ? 2 : 1;        // 2: the samples' input is staged in LDS (the fast form: Cin a multiple of 512, <= 144 KiB)
}

// epi 0: y = act ? lrelu(conv + bias, slope) : conv + bias, y2 (optional) = relu(y)
// epi 1: raw (optional) = bf16(conv + bias); stats[4][B][Cout] of raw per (sample, channel); y = lrelu(norm(raw), slope);
//        y2 (optional) = relu(norm(raw))                                    [InstanceNorm2d(affine=False, eps)]
// epi 2: g = conv (no bias).  Channels [0, bwd_c0): y2 = g (the plain data gradient).  Channels [bwd_c0, Cout), as channel
//        c - bwd_c0 of the normed tensor z / stats_in / g2 / y:  dzn = norm(z) > 0 ? g + g2 : slope * g;
//        y = invstd * (dzn - mean(dzn) - xhat * mean(dzn * xhat))          [the backward of epi 1 for that tensor]
extern "C" int s2s_convsm_nhwc(int dtype, int mode, const void* x, int ldx, int Cin, const void* w_packed, const float* bias,
                               int epi, int act, float slope, float eps, void* raw, int ldraw, void* y, int ldy, void* y2,
                               int ldy2, float* stats, const float* stats_in, const void* z, int ldz, const void* g2,
                               int ldg2, int bwd_c0, int B, int h, int w, int Cout, void* stream) {
  if (dtype != S2S_BF16) return S2S_ERR_DTYPE;
  if (!x || !w_packed) return S2S_ERR_NULL;
  SmallPlan pl;
  if (!small_plan(mode, B, h, w, Cin, Cout, &pl)) return S2S_ERR_SHAPE;
  if (epi < 0 || epi > 2 || (ldx % 8) || ldx < Cin) return S2S_ERR_SHAPE;
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)raw | (uintptr_t)y | (uintptr_t)y2 | (uintptr_t)z | (uintptr_t)g2) & 15)
    return S2S_ERR_ALIGN;
  const int HWo = mode == 2 ? 4 * h * w : h * w;
  if (epi == 1) {
    if (!y || !stats) return S2S_ERR_NULL;
    if (HWo <= 1) return S2S_ERR_SHAPE;           // torch raises for a single spatial element in training mode
    if ((ldy % 8) || (raw && (ldraw % 8)) || (y2 && (ldy2 % 8))) return S2S_ERR_SHAPE;
  } else if (epi == 0) {
    if (!y) return S2S_ERR_NULL;
    if ((ldy % 8) || (y2 && (ldy2 % 8))) return S2S_ERR_SHAPE;
  } else {
    if (bwd_c0 < 0 || bwd_c0 > Cout || (bwd_c0 % 16)) return S2S_ERR_SHAPE;
    if (bwd_c0 > 0 && (!y2 || (ldy2 % 8))) return S2S_ERR_NULL;
    if (bwd_c0 < Cout && (!y || !z || !stats_in || (ldy % 8) || (ldz % 8) || (g2 && (ldg2 % 8)))) return S2S_ERR_NULL;
    if (bwd_c0 < Cout && HWo <= 1) return S2S_ERR_SHAPE;
  }
  SmallArgs a{};
  a.x = (const bf16_t*)x; a.ldx = ldx; a.Cin = Cin;
  a.w = (const char*)w_packed; a.bias = bias;
  a.B = B; a.lgh = ilog2(h); a.lgw = ilog2(w); a.Cout = Cout;
  a.SG = pl.SG;
  a.nchunk = Cin / 32;
  if (mode == 1) {                                 // kernel rows that touch a real pixel for some output row
    a.t0y = h == 1 ? 1 : 0; a.nty = h == 1 ? 2 : 4;
    a.t0x = w == 1 ? 1 : 0; a.ntx = w == 1 ? 2 : 4;
  } else {
    a.t0y = 0; a.nty = h == 1 ? 1 : 2;
    a.t0x = 0; a.ntx = w == 1 ? 1 : 2;
  }
  a.lgntx = ilog2(a.ntx);
  a.epi = epi; a.act = act; a.slope = slope; a.eps = eps;
  a.raw = (bf16_t*)raw; a.ldraw = ldraw; a.y = (bf16_t*)y; a.ldy = ldy; a.y2 = (bf16_t*)y2; a.ldy2 = ldy2;
  a.stats = stats; a.stats_in = stats_in; a.z = (const bf16_t*)z; a.ldz = ldz; a.g2 = (const bf16_t*)g2; a.ldg2 = ldg2;
  a.bwd_c0 = bwd_c0;
  a.nx = cdiv(B, pl.SG); a.ny = Cout / 16;
  a.SP = pl.SP;
  hipStream_t s = (hipStream_t)stream;
  const int key = mode * 100 + pl.NB * 10 + pl.ldsx;
  switch (key) {
    case 111: return launch_small<1, 1, true>(a, pl.lds_bytes, s);
    case 121: return launch_small<1, 2, true>(a, pl.lds_bytes, s);
    case 110: return launch_small<1, 1, false>(a, pl.lds_bytes, s);
    case 120: return launch_small<1, 2, false>(a, pl.lds_bytes, s);
    case 140: return launch_small<1, 4, false>(a, pl.lds_bytes, s);
    case 211: return launch_small<2, 1, true>(a, pl.lds_bytes, s);
    case 221: return launch_small<2, 2, true>(a, pl.lds_bytes, s);
    case 210: return launch_small<2, 1, false>(a, pl.lds_bytes, s);
    default: return S2S_ERR_SHAPE;
  }
  return S2S_OK;
}
