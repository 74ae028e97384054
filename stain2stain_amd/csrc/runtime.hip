// Stream utilities of the C ABI (no kernels).
//
// s2s_stream_create_cu_mask: a HIP stream whose kernels are confined to a subset of the compute units.  The fused
// trainers put the MFMA-bound weight-gradient kernels on a side stream (engine.run_on_side); a weight-gradient
// workgroup pair takes every vector register of its CU, so the HBM-bound BatchNorm / up-sampling backward kernels of the
// main stream can only slip in where that kernel leaves a CU free.  Masking a quarter of the CUs out of the side stream
// reserves them for the main stream while both are busy.
#include <hip/hip_runtime.h>

#include "common.h"

extern "C" int s2s_stream_create_cu_mask(const unsigned* mask_words, int n_words, long* out_stream) {
  if (!mask_words || !out_stream) return S2S_ERR_NULL;
  if (n_words <= 0 || n_words > 64) return S2S_ERR_SHAPE;
  bool any = false;
  for (int i = 0; i < n_words; ++i) any = any || mask_words[i] != 0u;
  if (!any) return S2S_ERR_SHAPE;                      // a queue with no CU never runs
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words) != hipSuccess) {
    (void)hipGetLastError();
    return S2S_ERR_LAUNCH;
  }
  *out_stream = (long)(uintptr_t)s;
  return S2S_OK;
}

extern "C" int s2s_stream_destroy(void* stream) {
  if (!stream) return S2S_ERR_NULL;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? S2S_OK : S2S_ERR_LAUNCH;
}

// Timing-only events for the per-kernel brackets of bench.py (ops._timed).  A default HIP event performs a system-scope
// release fence when it completes (so that the host may read what the stream wrote); hipEventDisableSystemFence drops
// it for events that only carry a timestamp, which takes most of the marker's cost off the bracketed kernel.
extern "C" int s2s_event_create(int timing_only, long* out_event) {
  if (!out_event) return S2S_ERR_NULL;
  hipEvent_t e = nullptr;
  const unsigned flags = timing_only ? hipEventDisableSystemFence : hipEventDefault;
  if (hipEventCreateWithFlags(&e, flags) != hipSuccess) { (void)hipGetLastError(); return S2S_ERR_LAUNCH; }
  *out_event = (long)(uintptr_t)e;
  return S2S_OK;
}

extern "C" int s2s_event_record(void* event, void* stream) {
  if (!event) return S2S_ERR_NULL;
  return hipEventRecord((hipEvent_t)event, (hipStream_t)stream) == hipSuccess ? S2S_OK : S2S_ERR_LAUNCH;
}

// milliseconds between two recorded events (both complete: synchronise the stream or the later event first)
extern "C" int s2s_event_elapsed_ms(void* start, void* stop, float* out_ms) {
  if (!start || !stop || !out_ms) return S2S_ERR_NULL;
  return hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? S2S_OK : S2S_ERR_LAUNCH;
}

extern "C" int s2s_event_synchronize(void* event) {
  if (!event) return S2S_ERR_NULL;
  return hipEventSynchronize((hipEvent_t)event) == hipSuccess ? S2S_OK : S2S_ERR_LAUNCH;
}

extern "C" int s2s_event_destroy(void* event) {
  if (!event) return S2S_ERR_NULL;
  return hipEventDestroy((hipEvent_t)event) == hipSuccess ? S2S_OK : S2S_ERR_LAUNCH;
}
