// Paired tile preparation on the GPU: crop + flip (shared parameters for source and target) + to_tensor + Normalize.
//
// Replaces the per-sample CPU transform chain of the reference's PairedDataset.__getitem__
// (src/data/paired_data_module.py:170-199): RandomCrop.get_params -> TF.crop on both images with the SAME
// (i, j, h, w), shared horizontal / vertical flips, TF.to_tensor (uint8 HWC -> float CHW, x/255) and
// Normalize(mean 0.5, std 0.5) -> (x/255 - 0.5)/0.5.  Decoding (cv2/PIL) stays on the host; this kernel takes
// the decoded uint8 HWC RGB images resident in HBM and writes the NCHW fp32 tensors the model boundary expects.
// One thread per output pixel: 3-byte reads are contiguous across lanes, the three plane writes coalesced.
// Bit-exact with the fp32 expression above (IEEE division, same operation order).
#include "common.h"

namespace {

__global__ void paired_prepare_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ tgt,
                                      const int* __restrict__ params, float* __restrict__ out_src,
                                      float* __restrict__ out_tgt, int B, int Hs, int Ws, int S) {
  const long total = (long)B * S * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S);
    const int y = (int)((i / S) % S);
    const int n = (int)(i / ((long)S * S));
    const int top = params[n * 4 + 0], left = params[n * 4 + 1];
    const int hflip = params[n * 4 + 2], vflip = params[n * 4 + 3];
    const int sy = top + (vflip ? S - 1 - y : y);
    const int sx = left + (hflip ? S - 1 - x : x);
    const long ip = (((long)n * Hs + sy) * Ws + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long op = (((long)n * 3 + c) * S + y) * S + x;
      out_src[op] = ((float)src[ip + c] / 255.0f - 0.5f) / 0.5f;
      out_tgt[op] = ((float)tgt[ip + c] / 255.0f - 0.5f) / 0.5f;
    }
  }
}

}  // namespace

// src/tgt: uint8 [B][Hs][Ws][3] (decoded RGB, HWC); params: int32 [B][4] = {top, left, hflip, vflip};
// out_*: float [B][3][S][S].  Crop windows must lie inside the images (validated on the host by the caller's
// parameter generator; the kernel does not clamp).
extern "C" int s2s_paired_crop_flip_normalize(const void* src_u8, const void* tgt_u8, const int* params,
                                              float* out_src, float* out_tgt, int B, int Hs, int Ws, int S,
                                              void* stream) {
  if (!src_u8 || !tgt_u8 || !params || !out_src || !out_tgt) return S2S_ERR_NULL;
  if (B <= 0 || S <= 0 || Hs < S || Ws < S) return S2S_ERR_SHAPE;
  const long total = (long)B * S * S;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(paired_prepare_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src_u8, (const unsigned char*)tgt_u8, params, out_src, out_tgt, B, Hs, Ws,
                     S);
  S2S_LAUNCH_CHECK();
  return S2S_OK;
}
