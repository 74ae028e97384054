"""Tensor-level wrappers over the C ABI (include/stain2stain_hip.h).

Activations are torch tensors of logical shape [B, H, W, C] (NHWC) whose last dim is contiguous; a
channel slice of a wider NHWC buffer is a valid operand (its pixel stride is passed as ``ld``).
dtype bf16 -> throughput mode, float32 -> split-bf16 parity mode.  Every wrapper validates on the
host what the kernel assumes, launches on torch's current stream, and turns a non-zero status into
RuntimeError (the reference's only error convention is Python exceptions, src/utils/utils.py:65-80).
"""
from __future__ import annotations

import ctypes
import os
import sys
from typing import Optional, Tuple

import torch

from . import _native

BF16, F32 = 0, 1


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise RuntimeError(f"stain2stain_amd: unsupported activation dtype {t.dtype}")


# the current HIP stream's handle, ~80 times per optimisation step: torch.cuda.current_stream() builds a Stream object
# through four layers of device-index helpers (8 us a call, a quarter of the host time of a pix2pix step); the raw getter
# is one C call
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _quiesce_at_exit() -> None:
    """Before the interpreter tears anything down: destroy unreachable captured graphs (they live in reference cycles with
    their trainers) and let the device drain, while torch's and the HIP runtime's state are still complete."""
    import gc
    gc.collect()
    try:
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except Exception:  # noqa: BLE001 -- nothing useful can be done about a failing device at exit
        pass


import atexit as _atexit  # noqa: E402

_atexit.register(_quiesce_at_exit)


def _stream() -> int:
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _nhwc(t: torch.Tensor) -> Tuple[int, int]:
    """(data_ptr, pixel stride) of an NHWC view; raises if the view is not expressible as base+ld."""
    if not t.is_cuda:
        raise RuntimeError("stain2stain_amd: tensor is not on the GPU (there is no CPU path)")
    if t.dim() != 4 or t.stride(3) != 1:
        raise RuntimeError("stain2stain_amd: expected an NHWC view with contiguous channels")
    b, h, w, _ = t.shape
    ld = t.stride(2) if w > 1 else (t.stride(1) if h > 1 else (t.stride(0) if b > 1 else t.shape[3]))
    if w > 1 and h > 1 and t.stride(1) != w * ld:
        raise RuntimeError("stain2stain_amd: NHWC view has a row pitch")
    if b > 1 and t.stride(0) != h * w * ld:
        raise RuntimeError("stain2stain_amd: NHWC view has an image pitch")
    return t.data_ptr(), ld


def _f32(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise RuntimeError("stain2stain_amd: expected a contiguous fp32 GPU tensor")
    return t.data_ptr()


def _ptr(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if not t.is_cuda or not t.is_contiguous():
        raise RuntimeError("stain2stain_amd: expected a contiguous GPU tensor")
    return t.data_ptr()


def _L():
    return _native.lib()


# ------------------------------------------------------------------------------------------------
# persistent scratch for the split-K slabs
# ------------------------------------------------------------------------------------------------
# The weight-gradient kernels write tens of MB of fp32 partial slabs per launch.  Taking them from torch's caching
# allocator ties them to the stream that happens to be current (the allocator pools per stream): a step that runs its
# weight gradients on another stream than the previous one finds no cached block, falls through to hipMalloc and stalls
# the queue for ~0.2 ms per launch.  One grow-only buffer per (device, tag, stream) instead: launches on one stream are
# serialised, so consecutive launches may share it, and a buffer is only ever replaced (grown) by the stream that uses
# it -- the allocator returns the old block to that same stream's pool, where any re-use is ordered behind the launches
# that still read it.  (Keyed by (device, tag) alone, a larger launch on the side stream could retire a block the compute
# stream had allocated while the side stream's previous launch was still reading its slabs: ADVICE r2.)
_WORKSPACE = {}
# A captured training step bakes the buffer's address into its hipGraph, and a replay is not ordered against a later
# host-side replacement of the buffer: with one process-global buffer per (device, tag, stream), a second, wider trainer
# on the same stream would grow it -- returning the old block to the allocator -- while the first trainer's graph still
# writes its slabs there on every replay.  So the buffers belong to an OWNER: a trainer brackets its step
# with ``workspace_owner(self)``: nobody else can replace its buffers, and they are dropped with the trainer
# (``release_workspaces``).  Launches outside any trainer (the module path, tests) share the anonymous owner.
_WS_OWNER = 0


class workspace_owner:
    def __init__(self, owner):
        self.key = id(owner)

    def __enter__(self):
        global _WS_OWNER
        self.prev, _WS_OWNER = _WS_OWNER, self.key
        return self

    def __exit__(self, *exc):
        global _WS_OWNER
        _WS_OWNER = self.prev
        return False


def release_workspaces(owner) -> None:
    k = id(owner)
    for key in [key for key in _WORKSPACE if key[3] == k]:
        del _WORKSPACE[key]


def _workspace(device: torch.device, numel: int, tag: str) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag, _stream(), _WS_OWNER)
    buf = _WORKSPACE.get(key)
    if buf is None or buf.numel() < numel:
        # (inside a capture this happens for the capture stream only -- torch.cuda.graph captures on a stream of its own,
        #  the side stream's buffers exist since the eager warm-up step -- and the buffer, taken from the graph's private
        #  pool, stays referenced here until its owner releases it)
        buf = torch.empty((max(numel, 1 << 20),), dtype=torch.float32, device=device)
        _WORKSPACE[key] = buf
    return buf[:numel]


# ------------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream
# ------------------------------------------------------------------------------------------------
_PROFILE: Optional[list] = None
_PROFILE_ONLY: Optional[set] = None
PROFILE_TAG = ""          # set by the engines around a network's passes ("G" / "D"): bench.py attributes brackets by it


def profile_start(only=None) -> None:
    """Record HIP events around op launches; ``only`` restricts it to the named ops (cheaper: an event pair
    costs a few microseconds of host time per launch)."""
    global _PROFILE, _PROFILE_ONLY
    _PROFILE = []
    _PROFILE_ONLY = set(only) if only else None


def cu_masked_stream(device, spec: str) -> torch.cuda.Stream:
    """A HIP stream confined to a subset of the compute units: ``spec`` = "K:M" enables CU i iff i % M < K (e.g. "3:4" =
    three quarters of the chip, spread evenly whatever the CU numbering).  For the weight-gradient side stream."""
    k, m = (int(v) for v in spec.split(":"))
    if not 0 < k <= m:
        raise ValueError("CU mask spec K:M needs 0 < K <= M")
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), k, m)
    if key in _MASKED_STREAMS:
        return _MASKED_STREAMS[key]
    n = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (ctypes.c_uint32 * ((n + 31) // 32))()
    for i in range(n):
        if i % m < k:
            words[i // 32] |= 1 << (i % 32)
    out = ctypes.c_long(0)
    with torch.cuda.device(dev):
        _native.check(_L().s2s_stream_create_cu_mask(ctypes.addressof(words), len(words), ctypes.addressof(out)),
                      "stream_create_cu_mask")
    # The hipStream_t lives as long as the process (one per (device, K, M), cached): torch's caching allocator keys the
    # blocks that were allocated under `with torch.cuda.stream(side)` -- and their pending free-events -- by the raw
    # stream handle and keeps them after the Python object is gone, so destroying the stream when its ExternalStream
    # wrapper is collected leaves the allocator with a dangling handle that it touches again on a later free / at
    # teardown (the round-2 exit-time segmentation fault of the GPU suite).  ExternalStream never owned the handle;
    # nothing destroys it now, process teardown releases the queue.
    stream = torch.cuda.ExternalStream(out.value, device=dev)
    _MASKED_STREAMS[key] = stream
    return stream


_MASKED_STREAMS = {}


def side_stream_for(device) -> Optional[torch.cuda.Stream]:
    """The trainers' weight-gradient stream: None with S2S_WGRAD_STREAM=0, CU-masked with S2S_WGRAD_CUS=K:M."""
    import os
    if os.environ.get("S2S_WGRAD_STREAM", "1") == "0":
        return None
    spec = os.environ.get("S2S_WGRAD_CUS", "")
    if spec and spec != "0":
        return cu_masked_stream(device, spec)
    return torch.cuda.Stream(device=device)


_SHARED_SIDE = {}


def shared_side_stream(device) -> Optional[torch.cuda.Stream]:
    """One weight-gradient stream per device for the autograd Functions of the drop-in modules (the fused trainers own
    theirs): created once, None with S2S_WGRAD_STREAM=0."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SHARED_SIDE:
        _SHARED_SIDE[key] = side_stream_for(dev)
    return _SHARED_SIDE[key]


class TimingEvent:
    """A HIP event that only carries a timestamp (``s2s_event_create(timing_only=1)``: no system-scope fence when it
    completes), with the two methods of ``torch.cuda.Event`` the brackets use.  S2S_TIMING_EVENTS=0: torch's events."""

    def __init__(self):
        out = ctypes.c_long(0)
        _native.check(_L().s2s_event_create(1, ctypes.addressof(out)), "event_create")
        self._e = out.value

    def record(self) -> None:
        _native.check(_L().s2s_event_record(self._e, _stream()), "event_record")

    def elapsed_time(self, end: "TimingEvent") -> float:
        ms = ctypes.c_float(0.0)
        _native.check(_L().s2s_event_synchronize(end._e), "event_synchronize")
        _native.check(_L().s2s_event_elapsed_ms(self._e, end._e, ctypes.addressof(ms)), "event_elapsed")
        return float(ms.value)

    def __del__(self):
        try:
            if self._e and not sys.is_finalizing():
                _L().s2s_event_destroy(self._e)
        except Exception:  # noqa: BLE001
            pass


def _timing_event():
    if os.environ.get("S2S_TIMING_EVENTS", "1") == "0":
        return torch.cuda.Event(enable_timing=True)
    return TimingEvent()


def profile_stop() -> list:
    """[(op name, algorithmic work, start event, end event)] recorded since profile_start()."""
    global _PROFILE
    out, _PROFILE = _PROFILE or [], None
    return out


def _timed(name: str, work_fn=None):
    def deco(fn):
        def wrapped(*args, **kwargs):
            if _PROFILE is None or (_PROFILE_ONLY is not None and name not in _PROFILE_ONLY):
                return fn(*args, **kwargs)
            e0 = _timing_event()
            e1 = _timing_event()
            e0.record()
            out = fn(*args, **kwargs)
            e1.record()
            _PROFILE.append((name + ("@" + PROFILE_TAG if PROFILE_TAG else ""), work_fn(*args, **kwargs) if work_fn else 0.0, e0, e1))
            return out
        wrapped.__name__ = fn.__name__
        wrapped.__doc__ = fn.__doc__
        return wrapped
    return deco


def _conv_flops(x0, x1, w_packed, bias, cout, **kw) -> float:
    B, H, W, c0 = x0.shape
    cin = c0 + (x1.shape[3] if x1 is not None else 0)
    return 2.0 * B * H * W * cout * 9 * cin


def _wgrad_flops(dy, x0, x1, grad_oihw, accumulate=False) -> float:
    B, H, W, cout = dy.shape
    cin = x0.shape[3] + (x1.shape[3] if x1 is not None else 0)
    return 2.0 * B * H * W * cout * 9 * cin


# ------------------------------------------------------------------------------------------------
# convolutions
# ------------------------------------------------------------------------------------------------
@_timed("conv3x3_mfma", _conv_flops)
def conv3x3(x0: torch.Tensor, x1: Optional[torch.Tensor], w_packed: torch.Tensor, bias: Optional[torch.Tensor],
            cout: int, *, want_stats: bool = False, scale: Optional[torch.Tensor] = None,
            shift: Optional[torch.Tensor] = None, relu: bool = False, out: Optional[torch.Tensor] = None):
    """y[B,H,W,cout] = conv3x3(cat([x0, x1], C)); returns (y, stat_part or None)."""
    B, H, W, c0 = x0.shape
    dt = _dt(x0)
    p0, ld0 = _nhwc(x0)
    p1, ld1, c1 = 0, 8, 0
    if x1 is not None:
        if x1.shape[:3] != x0.shape[:3] or x1.dtype != x0.dtype:
            raise RuntimeError("stain2stain_amd: conv3x3 sources disagree in shape/dtype")
        p1, ld1 = _nhwc(x1)
        c1 = x1.shape[3]
    if w_packed.dtype != x0.dtype or w_packed.numel() != _L().s2s_pack_conv3x3_fwd_elems(cout, c0 + c1):
        raise RuntimeError("stain2stain_amd: packed weight does not match (cout, cin, dtype)")
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=x0.dtype, device=x0.device)
    elif out.shape != (B, H, W, cout) or out.dtype != x0.dtype:
        raise RuntimeError("stain2stain_amd: conv3x3 output buffer mismatch")
    py, ldy = _nhwc(out)
    stat, nb = None, 0
    if want_stats:
        nb = _L().s2s_conv3x3_stat_rows(dt, B, H, W, cout, c0, c1, ld0, ld1, int(bias is not None))
        _native.check(min(nb, 0), "conv3x3_stat_rows")
        stat = torch.empty((2, cout, nb), dtype=torch.float32, device=x0.device)
    kwork = None
    if not want_stats:                    # few output tiles (small batches): the chunk range is split over workgroups
        nsplit = _L().s2s_conv3x3_ksplit(dt, B, H, W, cout, c0 + c1)
        if nsplit > 1:
            kwork = torch.empty((nsplit, B * H * W, cout), dtype=torch.float32, device=x0.device)
    rc = _L().s2s_conv3x3_nhwc_s(dt, p0, ld0, c0, p1, ld1, c1, _ptr(w_packed), _f32(bias), py, ldy, _f32(stat), nb,
                                 _f32(scale), _f32(shift), int(relu), _f32(kwork), B, H, W, cout, _stream())
    _native.check(rc, "conv3x3")
    return out, stat


def conv3x3_wgrad(dy: torch.Tensor, x0: torch.Tensor, x1: Optional[torch.Tensor], grad_oihw: torch.Tensor,
                  accumulate: bool = False) -> None:
    B, H, W, cout = dy.shape
    c0 = x0.shape[3]
    dt = _dt(dy)
    if x0.dtype != dy.dtype or x0.shape[:3] != dy.shape[:3]:
        raise RuntimeError("stain2stain_amd: conv3x3_wgrad operand mismatch")
    pdy, lddy = _nhwc(dy)
    p0, ld0 = _nhwc(x0)
    p1, ld1, c1 = 0, 8, 0
    if x1 is not None:
        p1, ld1 = _nhwc(x1)
        c1 = x1.shape[3]
    if tuple(grad_oihw.shape) != (cout, c0 + c1, 3, 3):
        raise RuntimeError("stain2stain_amd: conv3x3_wgrad gradient buffer has the wrong shape")
    s = _L().s2s_conv3x3_wgrad_splits(dt, B, H, W, c0 + c1, cout)
    part = _workspace(dy.device, s * 9 * cout * (c0 + c1), "wgrad")
    if _PROFILE is not None and (_PROFILE_ONLY is None or "conv3x3_wgrad_mfma" in _PROFILE_ONLY):
        # event-bracketed: the split MFMA kernel and the fold of its slabs as two launches, each timed on its own
        # (rocprofv3 lists them as two kernels; the product path below issues them from one call)
        for name, phase, work in (("conv3x3_wgrad_mfma", 1, _wgrad_flops(dy, x0, x1, grad_oihw)), ("wgrad_fold", 2, 0.0)):
            e0 = _timing_event()
            e1 = _timing_event()
            e0.record()
            rc = _L().s2s_conv3x3_wgrad_phase(dt, pdy, lddy, cout, p0, ld0, c0, p1, ld1, c1, _f32(part), _f32(grad_oihw),
                                              int(accumulate), B, H, W, phase, _stream())
            e1.record()
            _native.check(rc, "conv3x3_wgrad")
            _PROFILE.append((name, work, e0, e1))
        return
    rc = _L().s2s_conv3x3_wgrad_nhwc(dt, pdy, lddy, cout, p0, ld0, c0, p1, ld1, c1, _f32(part), _f32(grad_oihw),
                                     int(accumulate), B, H, W, _stream())
    _native.check(rc, "conv3x3_wgrad")


@_timed("stem_fwd")
def stem_fwd(x_nchw: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype,
             want_stats: bool = True):
    B, cin, H, W = x_nchw.shape
    cout = w.shape[0]
    if tuple(w.shape) != (cout, cin, 3, 3):
        raise RuntimeError("stain2stain_amd: stem weight shape")
    y = torch.empty((B, H, W, cout), dtype=dtype, device=x_nchw.device)
    stat = None
    if want_stats:
        stat = torch.empty((2, cout, _L().s2s_stem_stat_blocks(B, H, W)), dtype=torch.float32, device=y.device)
    rc = _L().s2s_stem_conv3x3_fwd(_dt(y), _f32(x_nchw), _f32(w), _f32(bias), y.data_ptr(), cout, _f32(stat), B, H, W,
                                   cin, cout, _stream())
    _native.check(rc, "stem_conv3x3_fwd")
    return y, stat


@_timed("stem_wgrad")
def stem_wgrad(dy: torch.Tensor, x_nchw: torch.Tensor, dw: torch.Tensor, dbias: Optional[torch.Tensor],
               accumulate: bool = False) -> None:
    B, H, W, cout = dy.shape
    cin = x_nchw.shape[1]
    pdy, lddy = _nhwc(dy)
    nb = _L().s2s_stem_wgrad_blocks(B, H, W)
    part = torch.empty(((cin + 2) // 3, 2 * nb, cout, 32), dtype=torch.float32, device=dy.device)
    rc = _L().s2s_stem_conv3x3_wgrad(_dt(dy), pdy, lddy, _f32(x_nchw), _f32(part), _f32(dw), _f32(dbias),
                                     int(accumulate), B, H, W, cin, cout, _stream())
    _native.check(rc, "stem_conv3x3_wgrad")


@_timed("head_fwd")
def head_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    B, H, W, C = x.shape
    cout = w.shape[0]
    px, ldx = _nhwc(x)
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    rc = _L().s2s_head_conv1x1_fwd(_dt(x), px, ldx, _f32(w.reshape(cout, C)), _f32(bias), _f32(y), B, H, W, C, cout,
                                   _stream())
    _native.check(rc, "head_conv1x1_fwd")
    return y


@_timed("head_bwd")
def head_bwd(dy_nchw: torch.Tensor, x: torch.Tensor, w: torch.Tensor, dw: torch.Tensor, dbias: Optional[torch.Tensor],
             accumulate: bool = False) -> torch.Tensor:
    B, H, W, C = x.shape
    cout = w.shape[0]
    px, ldx = _nhwc(x)
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    nb = _L().s2s_head_wgrad_blocks(B, H, W)
    part = torch.empty((nb, cout, C + 1), dtype=torch.float32, device=x.device)
    rc = _L().s2s_head_conv1x1_bwd(_dt(x), _f32(dy_nchw), px, ldx, _f32(w.reshape(cout, C)), dx.data_ptr(), C,
                                   _f32(part), _f32(dw.reshape(cout, C)), _f32(dbias), int(accumulate), B, H, W, C,
                                   cout, _stream())
    _native.check(rc, "head_conv1x1_bwd")
    return dx


@_timed("head_loss_fused")
def head_loss_fused(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], u_nchw: torch.Tensor,
                    dw: torch.Tensor, dbias: Optional[torch.Tensor], want_v: bool = False, grad_scale: float = 1.0,
                    accumulate: bool = False):
    """Returns (loss, dx, v or None): head conv + MSE loss + both backward passes in one sweep."""
    B, H, W, C = x.shape
    cout = w.shape[0]
    px, ldx = _nhwc(x)
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    v = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device) if want_v else None
    nb = _L().s2s_head_loss_blocks(B, H, W)
    part = torch.empty((nb, cout, C + 1), dtype=torch.float32, device=x.device)
    lpart = torch.empty((nb,), dtype=torch.float64, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    rc = _L().s2s_head_loss_fused(_dt(x), px, ldx, _f32(w.reshape(cout, C)), _f32(bias), _f32(u_nchw), _f32(v),
                                  dx.data_ptr(), C, float(grad_scale), _f32(part), lpart.data_ptr(),
                                  _f32(dw.reshape(cout, C)), _f32(dbias), loss.data_ptr(), int(accumulate), B, H, W, C,
                                  cout, _stream())
    _native.check(rc, "head_loss_fused")
    return loss, dx, v


# ------------------------------------------------------------------------------------------------
# BatchNorm + ReLU (+ pool)
# ------------------------------------------------------------------------------------------------
@_timed("bn_finalize")
def bn_finalize(stat: torch.Tensor, count: int, gamma, beta, running_mean, running_var, num_batches,
                momentum: float = 0.1, eps: float = 1e-5, conv_bias: Optional[torch.Tensor] = None):
    """``conv_bias``: the statistics are those of the convolution WITHOUT its bias (which the caller did not add: it
    cancels behind a train-mode BatchNorm); the running mean is kept for conv + bias, as the reference's layers do."""
    _, C, nblk = stat.shape                  # channel-major partial sums [2][C][producer workgroups]
    dev = stat.device
    out = torch.empty((4, C), dtype=torch.float32, device=dev)  # mean, invstd, scale, shift
    rc = _L().s2s_bn_finalize_b(_f32(stat), nblk, C, count, _f32(gamma), _f32(beta), _f32(running_mean),
                                _f32(running_var), 0 if num_batches is None else num_batches.data_ptr(), momentum, eps,
                                out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                _f32(conv_bias), _stream())
    _native.check(rc, "bn_finalize")
    return out


def bn_partial_sums(stat: torch.Tensor) -> torch.Tensor:
    """Rank-local per-channel (sum, sum of squares) [2, C] of a conv epilogue's partial buffer: what SyncBatchNorm
    all-reduces; ``bn_finalize(sums.view(2, C, 1), global_count, ...)`` finishes the statistics."""
    _, C, nblk = stat.shape
    sums = torch.empty((2, C), dtype=torch.float32, device=stat.device)
    _native.check(_L().s2s_bn_partial_sums(_f32(stat), nblk, C, _f32(sums), _stream()), "bn_partial_sums")
    return sums


def bn_eval_prepare(gamma, beta, rmean, rvar, eps: float = 1e-5):
    C = gamma.numel()
    out = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    rc = _L().s2s_bn_eval_prepare(C, _f32(gamma), _f32(beta), _f32(rmean), _f32(rvar), eps, out[0].data_ptr(),
                                  out[1].data_ptr(), _stream())
    _native.check(rc, "bn_eval_prepare")
    return out


@_timed("bn_relu_apply")
def bn_relu_apply(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, out: Optional[torch.Tensor] = None,
                  pool: Optional[torch.Tensor] = None, want_pool: bool = False):
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    if out is None:
        out = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    py, ldy = _nhwc(out)
    pp, ldp = 0, 8
    if want_pool and pool is None:
        pool = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    if pool is not None:
        pp, ldp = _nhwc(pool)
    rc = _L().s2s_bn_relu_apply(_dt(x), px, ldx, _f32(scale), _f32(shift), py, ldy, pp, ldp, B, H, W, C, _stream())
    _native.check(rc, "bn_relu_apply")
    return out, pool


@_timed("maxpool2")
def maxpool2(x: torch.Tensor) -> torch.Tensor:
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    pool = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    rc = _L().s2s_maxpool2(_dt(x), px, ldx, pool.data_ptr(), C, B, H, W, C, _stream())
    _native.check(rc, "maxpool2")
    return pool


@_timed("bn_relu_bwd")
def bn_relu_bwd(g1: Optional[torch.Tensor], gp: Optional[torch.Tensor], x: torch.Tensor,
                stats: torch.Tensor, gamma: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
                dbias_conv: Optional[torch.Tensor], accumulate: bool = False, exchange=None,
                count_total: Optional[int] = None) -> torch.Tensor:
    """Returns dx (gradient wrt the conv output x).  stats = bn_finalize() output (rows mean, invstd, scale,
    shift); the ReLU mask / pool winner are recomputed from x with the forward's scale and shift.
    SyncBatchNorm: ``exchange(t)`` all-reduces (sum) the [2C] tensor of per-channel means between the reduction and the
    apply pass, ``count_total`` is the global element count they are taken over."""
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    p1, ld1 = (0, 8) if g1 is None else _nhwc(g1)
    p2, ld2 = (0, 8) if gp is None else _nhwc(gp)
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    nb = _L().s2s_bn_bwd_blocks(B, H, W, C)
    work = torch.empty((4 * nb * C + 2 * C,), dtype=torch.float32, device=x.device)
    if exchange is not None:
        for phase in (1, 2):
            rc = _L().s2s_bn_relu_bwd_phase(_dt(x), p1, ld1, p2, ld2, stats[2].data_ptr(), stats[3].data_ptr(), px, ldx,
                                            stats[0].data_ptr(), stats[1].data_ptr(),
                                            _f32(gamma), _f32(dgamma), _f32(dbeta), _f32(dbias_conv), int(accumulate),
                                            dx.data_ptr(), C, _f32(work), B, H, W, C,
                                            int(count_total if count_total is not None else B * H * W), phase, _stream())
            _native.check(rc, "bn_relu_bwd")
            if phase == 1:
                exchange(work[4 * nb * C:])
        return dx
    rc = _L().s2s_bn_relu_bwd(_dt(x), p1, ld1, p2, ld2, stats[2].data_ptr(), stats[3].data_ptr(), px, ldx,
                              stats[0].data_ptr(), stats[1].data_ptr(),
                              _f32(gamma), _f32(dgamma), _f32(dbeta), _f32(dbias_conv), int(accumulate),
                              dx.data_ptr(), C, _f32(work), B, H, W, C, _stream())
    _native.check(rc, "bn_relu_bwd")
    return dx


# ------------------------------------------------------------------------------------------------
# resampling
# ------------------------------------------------------------------------------------------------
@_timed("upsample2x_fwd")
def upsample2x_fwd(x: torch.Tensor, out: torch.Tensor, bias_nc: Optional[torch.Tensor] = None) -> None:
    B, Hin, Win, C = x.shape
    _, Hout, Wout, Co = out.shape
    if Co != C or out.dtype != x.dtype:
        raise RuntimeError("stain2stain_amd: upsample output slice mismatch")
    px, ldx = _nhwc(x)
    py, ldy = _nhwc(out)
    rc = _L().s2s_upsample2x_bilinear_ac_fwd(_dt(x), px, ldx, _f32(bias_nc), py, ldy, B, Hin, Win, Hout, Wout, C,
                                             _stream())
    _native.check(rc, "upsample2x_fwd")


@_timed("upsample2x_bwd")
def upsample2x_bwd(dy: torch.Tensor, hin: int, win: int) -> torch.Tensor:
    B, Hout, Wout, C = dy.shape
    pdy, lddy = _nhwc(dy)
    dx = torch.empty((B, hin, win, C), dtype=dy.dtype, device=dy.device)
    rc = _L().s2s_upsample2x_bilinear_ac_bwd(_dt(dy), pdy, lddy, dx.data_ptr(), C, B, hin, win, Hout, Wout, C,
                                             _stream())
    _native.check(rc, "upsample2x_bwd")
    return dx


def pixel_sum(x: torch.Tensor) -> torch.Tensor:
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    rc = _L().s2s_pixel_sum(_dt(x), px, ldx, _f32(out), B, H * W, C, 0, _stream())
    _native.check(rc, "pixel_sum")
    return out


# ------------------------------------------------------------------------------------------------
# flow-matching glue
# ------------------------------------------------------------------------------------------------
def time_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    t = t.reshape(-1).to(torch.float32).contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.float32, device=t.device)
    _native.check(_L().s2s_time_embedding(_f32(t), _f32(out), t.numel(), dim, _stream()), "time_embedding")
    return out


def linear_fwd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    B, K = x.shape
    N = w.shape[0]
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _native.check(_L().s2s_linear_fwd(_f32(x), _f32(w), _f32(b), _f32(y), B, K, N, _stream()), "linear_fwd")
    return y


def linear_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor],
               need_dx: bool = True, accumulate: bool = False) -> Optional[torch.Tensor]:
    B, K = x.shape
    N = w.shape[0]
    dx = torch.empty((B, K), dtype=torch.float32, device=x.device) if need_dx else None
    _native.check(_L().s2s_linear_bwd(_f32(dy), _f32(x), _f32(w), _f32(dx), _f32(dw), _f32(db), int(accumulate), B, K,
                                      N, _stream()), "linear_bwd")
    return dx


def silu_fwd(h: torch.Tensor) -> torch.Tensor:
    a = torch.empty_like(h)
    _native.check(_L().s2s_silu_fwd(_f32(h), _f32(a), h.numel(), _stream()), "silu_fwd")
    return a


def silu_bwd(h: torch.Tensor, da: torch.Tensor) -> torch.Tensor:
    dh = torch.empty_like(h)
    _native.check(_L().s2s_silu_bwd(_f32(h), _f32(da), _f32(dh), h.numel(), _stream()), "silu_bwd")
    return dh


@_timed("cfm_sample")
def cfm_sample(x0: torch.Tensor, x1: torch.Tensor, t: torch.Tensor, sigma: float = 0.0,
               eps: Optional[torch.Tensor] = None):
    xt = torch.empty_like(x0)
    ut = torch.empty_like(x0)
    B = x0.shape[0]
    _native.check(_L().s2s_cfm_sample(_f32(x0), _f32(x1), _f32(t), _f32(eps), float(sigma), _f32(xt), _f32(ut), B,
                                      x0.numel() // B, _stream()), "cfm_sample")
    return xt, ut


@_timed("mse_loss")
def mse_loss(v: torch.Tensor, u: torch.Tensor, want_grad: bool = True, grad_scale: float = 1.0):
    loss = torch.empty((), dtype=torch.float32, device=v.device)
    dv = torch.empty_like(v) if want_grad else None
    work = torch.empty((1024,), dtype=torch.float64, device=v.device)
    _native.check(_L().s2s_mse_loss(_f32(v), _f32(u), _f32(dv), float(grad_scale), loss.data_ptr(), work.data_ptr(),
                                    v.numel(), _stream()), "mse_loss")
    return loss, dv


def seg_loss(z: torch.Tensor, g: torch.Tensor, smooth: float = 1.0, dice_weight: float = 0.5, want_grad: bool = True,
             grad_scale: float = 1.0):
    """Dice + BCE-with-logits on a logit map; returns (out[3] = seg, dice, bce; dz or None)."""
    z = z.contiguous()
    g = g.to(torch.float32).contiguous()
    out = torch.empty((3,), dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z) if want_grad else None
    work = torch.empty((512 * 4 + 4,), dtype=torch.float64, device=z.device)
    _native.check(_L().s2s_seg_loss(_f32(z), _f32(g), _f32(dz), _f32(out), work.data_ptr(), z.numel(), float(smooth),
                                    float(dice_weight), float(grad_scale), _stream()), "seg_loss")
    return out, dz


def seg_loss_multiclass(z: torch.Tensor, target: torch.Tensor, ignore_index: int = -100, smooth: float = 1.0,
                        dice_weight: float = 0.5, want_grad: bool = True, grad_scale: float = 1.0,
                        validate: bool = True):
    """Softmax Dice (mean over classes) + cross entropy on [B,C,H,W] logits and [B,H,W] class indices;
    returns (out[3] = seg, dice, ce; dz or None).  ``validate`` reproduces the reference's failure on labels
    outside [0, C) (F.one_hot raises there) at the cost of one host sync."""
    if z.dim() != 4 or target.dim() != 3 or target.shape != (z.shape[0], z.shape[2], z.shape[3]):
        raise RuntimeError("stain2stain_amd: seg_loss_multiclass wants logits [B,C,H,W] and target [B,H,W]")
    B, C, H, W = z.shape
    z = z.contiguous()
    target = target.to(torch.int64).contiguous()
    if validate:
        lo, hi = int(target.min()), int(target.max())
        if lo < 0 or hi >= C:
            raise RuntimeError(f"stain2stain_amd: class index out of range [0, {C}) in segmentation target "
                               f"(min {lo}, max {hi})")
    out = torch.empty((3,), dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z) if want_grad else None
    work = torch.empty((513 * 26,), dtype=torch.float64, device=z.device)
    _native.check(_L().s2s_seg_loss_multiclass(_f32(z), target.data_ptr(), _f32(dz), _f32(out), work.data_ptr(), B,
                                               H * W, C, int(ignore_index), float(smooth), float(dice_weight),
                                               float(grad_scale), _stream()), "seg_loss_multiclass")
    return out, dz


def _mask_bhw(mask: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    B, C, H, W = like.shape
    if mask.dim() == 4 and mask.shape[1] == 1:
        mask = mask[:, 0]
    if tuple(mask.shape) != (B, H, W):
        raise RuntimeError(f"stain2stain_amd: mask shape {tuple(mask.shape)} does not match images {tuple(like.shape)}")
    return mask.to(torch.float32).contiguous()


def weighted_mse(v: torch.Tensor, u: torch.Tensor, mask: torch.Tensor, roi_lambda: float = 10.0,
                 want_grad: bool = True, grad_scale: float = 1.0):
    """sum(w (v-u)^2) / (sum(w) + 1e-8) with w = 1 + roi_lambda * mask broadcast over channels; (loss[1], dv)."""
    if v.dim() != 4 or v.shape != u.shape:
        raise RuntimeError("stain2stain_amd: weighted_mse wants v, u of equal shape [B,C,H,W]")
    B, C, H, W = v.shape
    v, u, m = v.contiguous(), u.contiguous(), _mask_bhw(mask, v)
    out = torch.empty((1,), dtype=torch.float32, device=v.device)
    dv = torch.empty_like(v) if want_grad else None
    work = torch.empty((512 * 2 + 2,), dtype=torch.float64, device=v.device)
    _native.check(_L().s2s_weighted_mse(_f32(v), _f32(u), _f32(m), _f32(dv), _f32(out), work.data_ptr(), B, C, H * W,
                                        float(roi_lambda), float(grad_scale), _stream()), "weighted_mse")
    return out, dv


def charbonnier_roi(pred: torch.Tensor, truth: torch.Tensor, mask: torch.Tensor, eps_charb: float = 1e-3,
                    eps_area: float = 1e-8) -> torch.Tensor:
    """sum(sqrt((pred-truth)^2 + eps_charb^2) * mask) / (sum(mask) * C + eps_area); value only."""
    if pred.dim() != 4 or pred.shape != truth.shape:
        raise RuntimeError("stain2stain_amd: charbonnier_roi wants pred, truth of equal shape [B,C,H,W]")
    B, C, H, W = pred.shape
    pred, truth, m = pred.contiguous(), truth.contiguous(), _mask_bhw(mask, pred)
    out = torch.empty((1,), dtype=torch.float32, device=pred.device)
    work = torch.empty((512 * 2 + 2,), dtype=torch.float64, device=pred.device)
    _native.check(_L().s2s_charbonnier_roi(_f32(pred), _f32(truth), _f32(m), _f32(out), work.data_ptr(), B, C, H * W,
                                           float(eps_charb), float(eps_area), _stream()), "charbonnier_roi")
    return out


def class_embed_add(temb: torch.Tensor, table: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    B, dim = temb.shape
    if table.shape[1] != dim or y.shape != (B,):
        raise RuntimeError("stain2stain_amd: class_embed_add shape mismatch")
    out = torch.empty_like(temb)
    _native.check(_L().s2s_class_embed_add(_f32(temb.contiguous()), _f32(table), y.data_ptr(), _f32(out), B, dim,
                                           _stream()), "class_embed_add")
    return out


def class_embed_bwd(dout: torch.Tensor, y: torch.Tensor, num_classes: int) -> torch.Tensor:
    B, dim = dout.shape
    dtable = torch.empty((num_classes, dim), dtype=torch.float32, device=dout.device)
    _native.check(_L().s2s_class_embed_bwd(_f32(dout.contiguous()), y.data_ptr(), _f32(dtable), 0, B, dim,
                                           num_classes, _stream()), "class_embed_bwd")
    return dtable


def ode_error_norm(e: torch.Tensor, y0: torch.Tensor, y1: torch.Tensor, atol: float, rtol: float) -> torch.Tensor:
    out = torch.empty((1,), dtype=torch.float32, device=e.device)
    work = torch.empty((1024,), dtype=torch.float64, device=e.device)
    _native.check(_L().s2s_ode_error_norm(_f32(e), _f32(y0), _f32(y1), float(atol), float(rtol), _f32(out),
                                          work.data_ptr(), e.numel(), _stream()), "ode_error_norm")
    return out


def axpy_(x: torch.Tensor, y: torch.Tensor, a: float) -> None:
    _native.check(_L().s2s_axpy(_f32(x), _f32(y), float(a), x.numel(), _stream()), "axpy")


def euler_tick_(t: torch.Tensor, table: torch.Tensor, counter: torch.Tensor) -> None:
    """t[:] = table[counter]; counter += 1 (device side; see s2s_euler_tick)."""
    if counter.dtype != torch.int32 or table.dtype != torch.float32:
        raise RuntimeError("stain2stain_amd: euler_tick wants an int32 counter and a float32 table")
    _native.check(_L().s2s_euler_tick(_f32(t), t.numel(), _f32(table), counter.data_ptr(), _stream()), "euler_tick")


def fill_(x: torch.Tensor, v: float) -> None:
    _native.check(_L().s2s_fill_f32(_f32(x), float(v), x.numel(), _stream()), "fill")


# ------------------------------------------------------------------------------------------------
# optimiser / packing / layout
# ------------------------------------------------------------------------------------------------
@_timed("adam_step_")
def adam_step_(p, g, m, v, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
               weight_decay: float = 0.0, grad_scale: float = 1.0) -> None:
    _native.check(_L().s2s_adam_step(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), int(step), float(lr), float(beta1),
                                     float(beta2), float(eps), float(weight_decay), float(grad_scale), _stream()),
                  "adam_step")


def adam_hyper(step: int, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
               grad_scale: float) -> list:
    """The eight floats ``adam_step_dev_`` reads, formed exactly as s2s_adam_step forms its kernel arguments."""
    import math
    import struct
    f32 = lambda x: struct.unpack("f", struct.pack("f", float(x)))[0]      # the C entry point takes beta as `float`
    bc1 = 1.0 - math.pow(f32(beta1), float(step))
    bc2 = 1.0 - math.pow(f32(beta2), float(step))
    return [float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), bc1, math.sqrt(bc2), float(grad_scale)]


class capture_graph:
    """``torch.cuda.graph(g)`` with Python's cyclic collector out of the way.  A collection that runs DURING a capture can
    destroy an earlier, unreachable ``torch.cuda.CUDAGraph`` (e.g. the sampler's GraphedVelocity, which forms a cycle with
    its network): hipGraphExecDestroy / the release of its private memory pool inside another stream capture aborts the
    process (seen on ROCm 7 / torch 2.10, whose ``torch.cuda.graph`` no longer collects on entry).  So: collect first,
    keep the collector off while capturing, restore it afterwards.

    Capture mode ``thread_local``: with a process group alive, ProcessGroupNCCL's watchdog THREAD polls the events of
    earlier (eager) collectives with hipEventQuery; under the default ``global`` mode any such call from another thread
    while this thread captures is an error, which the watchdog turns into an abort of the process (seen in the full GPU
    suite: ``bench.py`` under S2S_FORCE_DDP=1 died in WorkNCCL::finishedGPUExecutionInternal when the graph leg's capture
    raced the watchdog's poll of the warm-up steps' all-reduces)."""

    def __init__(self, graph: "torch.cuda.CUDAGraph"):
        self._ctx = torch.cuda.graph(graph, capture_error_mode="thread_local")

    def __enter__(self):
        import gc
        self._gc = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            torch.cuda.synchronize()
            return self._ctx.__enter__()
        except BaseException:          # a failed entry never reaches __exit__: put the collector back here
            if self._gc:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self._ctx.__exit__(*exc)
        finally:
            if self._gc:
                gc.enable()


class AdamHyperRing:
    """The eight Adam scalars of a captured training step: ``dev`` is what the replayed kernel reads; ``push`` refreshes
    it from the host in stream order without a synchronisation (pinned staging slots, each re-used only after the copy
    that read it has completed -- the host runs ahead of the GPU by a step or two, never by the ring's length without
    this check stalling it)."""
    SLOTS = 16

    def __init__(self, device):
        self.dev = torch.zeros(8, dtype=torch.float32, device=device)
        self._host = torch.empty((self.SLOTS, 8), dtype=torch.float32).pin_memory()
        self._done = [None] * self.SLOTS
        self._i = 0

    def push(self, values) -> None:
        i = self._i
        self._i = (i + 1) % self.SLOTS
        if self._done[i] is not None:
            self._done[i].synchronize()
        self._host[i] = torch.tensor(values, dtype=torch.float32)
        self.dev.copy_(self._host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._done[i] = ev


def adam_step_dev_(p, g, m, v, hyper: torch.Tensor) -> None:
    """Adam with its scalars in a device float[8] (``adam_hyper``): what a captured training step replays."""
    if hyper.numel() != 8:
        raise RuntimeError("stain2stain_amd: adam hyper-parameter buffer must hold 8 floats")
    _native.check(_L().s2s_adam_step_dev(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), _f32(hyper), _stream()),
                  "adam_step_dev")


@_timed("pack_conv3x3")
def pack_conv3x3(w_oihw: torch.Tensor, dtype: torch.dtype, want_dgrad: bool = True, out=None):
    cout, cin = w_oihw.shape[:2]
    dev = w_oihw.device
    if out is None:
        wf = torch.empty((_L().s2s_pack_conv3x3_fwd_elems(cout, cin),), dtype=dtype, device=dev)
        wd = torch.empty((_L().s2s_pack_conv3x3_dgrad_elems(cout, cin),), dtype=dtype, device=dev) if want_dgrad else None
    else:
        wf, wd = out
    rc = _L().s2s_pack_conv3x3(_dt(wf), _f32(w_oihw), wf.data_ptr(), 0 if wd is None else wd.data_ptr(), cout, cin,
                               _stream())
    _native.check(rc, "pack_conv3x3")
    return wf, wd


@_timed("pack_conv3x3")
def pack_conv3x3_batched(desc: torch.Tensor, total: int, dtype: torch.dtype) -> None:
    """desc: int64 [nlayers, 6] device tensor {w ptr, wf ptr, wd ptr, Cout, Cin, first tile}; total = tiles."""
    code = BF16 if dtype == torch.bfloat16 else F32
    if not desc.is_cuda:
        raise RuntimeError("stain2stain_amd: the layer descriptor table must live on the GPU")
    _native.check(_L().s2s_pack_conv3x3_batched(code, desc.data_ptr(), desc.shape[0], int(total), _stream()),
                  "pack_conv3x3_batched")


def nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty((B, H, W, C), dtype=dtype, device=x.device)
    py, ldy = _nhwc(out)
    _native.check(_L().s2s_nchw_to_nhwc(_dt(out), _f32(x), py, ldy, B, C, H, W, _stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _native.check(_L().s2s_nhwc_to_nchw(_dt(x), px, ldx, _f32(out), int(accumulate), B, C, H, W, _stream()),
                  "nhwc_to_nchw")
    return out


# ------------------------------------------------------------------------------------------------
# InstanceNorm2d + LeakyReLU (row a13: pix2pix generator / PatchGAN discriminator building block)
# ------------------------------------------------------------------------------------------------
@_timed("instnorm_lrelu_fwd")
def instnorm_lrelu_fwd(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                       eps: float = 1e-5, slope: float = 0.2, out: Optional[torch.Tensor] = None):
    """y = leaky_relu(instance_norm(x), slope) on an NHWC view; returns (y, stats[4][B][C])."""
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    y = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device) if out is None else out
    py, ldy = _nhwc(y)
    nb = _L().s2s_instnorm_blocks(B, H, W, C)
    _native.check(min(nb, 0), "instnorm_blocks")
    work = torch.empty(2 * B * C * nb, dtype=torch.float32, device=x.device)
    stats = torch.empty((4, B, C), dtype=torch.float32, device=x.device)
    rc = _L().s2s_instnorm_lrelu_fwd(_dt(x), px, ldx, _f32(gamma), _f32(beta), py, ldy, _f32(work), _f32(stats), B, H, W,
                                     C, eps, slope, _stream())
    _native.check(rc, "instnorm_lrelu_fwd")
    return y, stats


@_timed("instnorm_lrelu_bwd")
def instnorm_lrelu_bwd(g: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, slope: float = 0.2,
                       dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                       accumulate: bool = False) -> torch.Tensor:
    B, H, W, C = x.shape
    if g.shape != x.shape or g.dtype != x.dtype:
        raise RuntimeError("stain2stain_amd: instnorm_lrelu_bwd operand mismatch")
    pg, ldg = _nhwc(g)
    px, ldx = _nhwc(x)
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    nb = _L().s2s_instnorm_blocks(B, H, W, C)
    _native.check(min(nb, 0), "instnorm_blocks")
    work = torch.empty(2 * B * C * nb + 2 * B * C, dtype=torch.float32, device=x.device)
    rc = _L().s2s_instnorm_lrelu_bwd(_dt(x), pg, ldg, px, ldx, _f32(stats), dx.data_ptr(), C, _f32(dgamma), _f32(dbeta),
                                     int(accumulate), _f32(work), B, H, W, C, slope, _stream())
    _native.check(rc, "instnorm_lrelu_bwd")
    return dx


@_timed("conv2x2_mfma", lambda x, w_packed, bias, cout, pad, **kw: 2.0 * x.shape[0] * (x.shape[1] + (1 if pad else -1))
        * (x.shape[2] + (1 if pad else -1)) * cout * 4 * x.shape[3])
def conv2x2(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, pad: int, *,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """2x2-tap convolution on the MFMA loop (row a13): pad 0 -> output (Hi-1) x (Wi-1) ("valid"), pad 1 -> output
    (Hi+1) x (Wi+1) (the data-gradient / transposed form).  x: NHWC bf16 view; w_packed: bf16 [cin/32][4][cout][32]."""
    B, Hi, Wi, cin = x.shape
    H, W = (Hi + 1, Wi + 1) if pad else (Hi - 1, Wi - 1)
    if x.dtype != torch.bfloat16 or w_packed.dtype != torch.bfloat16:
        raise RuntimeError("stain2stain_amd: conv2x2 runs in bf16 only")
    if w_packed.numel() != ((cin + 31) // 32) * 4 * cout * 32:
        raise RuntimeError("stain2stain_amd: conv2x2 packed weight has the wrong size")
    px, ldx = _nhwc(x)
    y = torch.empty((B, H, W, cout), dtype=x.dtype, device=x.device) if out is None else out
    py, ldy = _nhwc(y)
    rc = _L().s2s_conv2x2_nhwc(_dt(x), px, ldx, cin, _ptr(w_packed), _f32(bias), py, ldy, 0, B, H, W, cout, int(pad),
                               _stream())
    _native.check(rc, "conv2x2")
    return y


@_timed("conv2x2_wgrad_mfma")
def conv2x2_wgrad(dy: torch.Tensor, x: torch.Tensor, grad2: torch.Tensor, accumulate: bool = False) -> None:
    """grad2[4][cout][cin] (+)= weight gradient of conv2x2(pad 0): dy [B,H,W,cout], x [B,H+1,W+1,cin], NHWC bf16."""
    B, H, W, cout = dy.shape
    cin = x.shape[3]
    if x.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16 or tuple(x.shape[:3]) != (B, H + 1, W + 1):
        raise RuntimeError("stain2stain_amd: conv2x2_wgrad operand mismatch")
    if tuple(grad2.shape) != (4, cout, cin):
        raise RuntimeError("stain2stain_amd: conv2x2_wgrad gradient buffer has the wrong shape")
    pdy, lddy = _nhwc(dy)
    px, ldx = _nhwc(x)
    s = _L().s2s_conv2x2_wgrad_splits(B, H, W, cin, cout)
    _native.check(min(s, 0), "conv2x2_wgrad_splits")
    part = torch.empty((s, 4, cout, cin), dtype=torch.float32, device=dy.device)
    rc = _L().s2s_conv2x2_wgrad_nhwc(_dt(dy), pdy, lddy, cout, px, ldx, cin, _f32(part), _f32(grad2), int(accumulate),
                                     B, H, W, _stream())
    _native.check(rc, "conv2x2_wgrad")


@_timed("conv4x4s1_mfma", lambda x, w_packed, bias, cout, pad, **kw: 2.0 * x.shape[0] * (x.shape[1] + (1 if pad == 2 else -1))
        * (x.shape[2] + (1 if pad == 2 else -1)) * cout * 16 * x.shape[3])
def conv4x4s1(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, pad: int) -> torch.Tensor:
    """4x4 stride-1 convolution on the MFMA loop (row a13, PatchGAN): pad 1 -> output (Hi-1) x (Wi-1)
    (nn.Conv2d(k=4, s=1, p=1)); pad 2 -> output (Hi+1) x (Wi+1) (its data gradient, flipped packing).
    x: NHWC bf16 view; w_packed: bf16 [cin/32][16][cout][32]."""
    B, Hi, Wi, cin = x.shape
    if pad not in (1, 2):
        raise RuntimeError("stain2stain_amd: conv4x4s1 pad must be 1 or 2")
    H, W = (Hi + 1, Wi + 1) if pad == 2 else (Hi - 1, Wi - 1)
    if x.dtype != torch.bfloat16 or w_packed.dtype != torch.bfloat16:
        raise RuntimeError("stain2stain_amd: conv4x4s1 runs in bf16 only")
    if w_packed.numel() != ((cin + 31) // 32) * 16 * cout * 32:
        raise RuntimeError("stain2stain_amd: conv4x4s1 packed weight has the wrong size")
    px, ldx = _nhwc(x)
    y = torch.empty((B, H, W, cout), dtype=x.dtype, device=x.device)
    rc = _L().s2s_conv4x4s1_nhwc(_dt(x), px, ldx, cin, _ptr(w_packed), _f32(bias), y.data_ptr(), cout, 0, B, H, W, cout,
                                 int(pad), _stream())
    _native.check(rc, "conv4x4s1")
    return y


@_timed("conv4x4s1_wgrad_mfma")
def conv4x4s1_wgrad(dy: torch.Tensor, x: torch.Tensor, grad16: torch.Tensor, accumulate: bool = False) -> None:
    """grad16[16][cout][cin] (+)= weight gradient of conv4x4s1(pad 1): dy [B,H,W,cout], x [B,H+1,W+1,cin], NHWC bf16."""
    B, H, W, cout = dy.shape
    cin = x.shape[3]
    if x.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16 or tuple(x.shape[:3]) != (B, H + 1, W + 1):
        raise RuntimeError("stain2stain_amd: conv4x4s1_wgrad operand mismatch")
    if tuple(grad16.shape) != (16, cout, cin):
        raise RuntimeError("stain2stain_amd: conv4x4s1_wgrad gradient buffer has the wrong shape")
    pdy, lddy = _nhwc(dy)
    px, ldx = _nhwc(x)
    s = _L().s2s_conv4x4s1_wgrad_splits(B, H, W, cin, cout)
    _native.check(min(s, 0), "conv4x4s1_wgrad_splits")
    part = torch.empty((s, 16, cout, cin), dtype=torch.float32, device=dy.device)
    rc = _L().s2s_conv4x4s1_wgrad_nhwc(_dt(dy), pdy, lddy, cout, px, ldx, cin, _f32(part), _f32(grad16), int(accumulate),
                                       B, H, W, _stream())
    _native.check(rc, "conv4x4s1_wgrad")


def space_to_depth_pad1(x: torch.Tensor) -> torch.Tensor:
    """xs[n,p,q,(r*2+s)*C+c] = xpad[n,2p+r,2q+s,c] (one-pixel zero border): NHWC bf16 [B,H,W,C] -> [B,H/2+1,W/2+1,4C]."""
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    xs = torch.empty((B, H // 2 + 1, W // 2 + 1, 4 * C), dtype=x.dtype, device=x.device)
    rc = _L().s2s_space_to_depth_pad1(_dt(x), px, ldx, xs.data_ptr(), 4 * C, 0, B, H, W, C, _stream())
    _native.check(rc, "space_to_depth_pad1")
    return xs


def depth_to_space_unpad1(xs: torch.Tensor) -> torch.Tensor:
    """Inverse of space_to_depth_pad1 (the border cells are dropped): [B,H/2+1,W/2+1,4C] -> [B,H,W,C]."""
    B, Hs, Ws, C4 = xs.shape
    H, W, C = 2 * (Hs - 1), 2 * (Ws - 1), C4 // 4
    pxs, ldxs = _nhwc(xs)
    x = torch.empty((B, H, W, C), dtype=xs.dtype, device=xs.device)
    rc = _L().s2s_space_to_depth_pad1(_dt(xs), x.data_ptr(), C, pxs, ldxs, 1, B, H, W, C, _stream())
    _native.check(rc, "depth_to_space_unpad1")
    return x


def pack_conv4x4(w: torch.Tensor, stride: int):
    """fp32 master [Cout,Cin,4,4] -> (forward operand, data-gradient operand) in bf16 for the 4x4 stride-1 / -2 kernels."""
    cout, cin = w.shape[:2]
    taps, K = (4, 4 * cin) if stride == 2 else (16, cin)
    wf = torch.empty(((K + 31) // 32, taps, cout, 32), dtype=torch.bfloat16, device=w.device)
    wd = torch.empty(((cout + 31) // 32, taps, K, 32), dtype=torch.bfloat16, device=w.device)
    rc = _L().s2s_pack_conv4x4(_f32(w.detach().contiguous()), wf.data_ptr(), wd.data_ptr(), cout, cin, stride, _stream())
    _native.check(rc, "pack_conv4x4")
    return wf, wd


def channel_sum(x: torch.Tensor) -> torch.Tensor:
    """fp32 [C] = sum over every pixel of every sample of an NHWC (bf16 / fp32) tensor."""
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    npix = B * H * W
    nb = _L().s2s_channel_sum_blocks(npix, C)
    _native.check(min(nb, 0), "channel_sum_blocks")
    work = torch.empty(2 * C * nb, dtype=torch.float32, device=x.device)
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    rc = _L().s2s_channel_sum(_dt(x), px, ldx, _f32(work), _f32(out), npix, C, 0, _stream())
    _native.check(rc, "channel_sum")
    return out


# ------------------------------------------------------------------------------------------------
# pix2pix G + D step (row a13): general 4x4-layer kernels in both dtypes, fused elementwise / loss kernels
# ------------------------------------------------------------------------------------------------
def _kxk_flops(x, w_packed, bias, cout, ks, pad, **kw) -> float:
    B, Hi, Wi, cin = x.shape
    H, W = Hi - (ks - 1) + 2 * pad, Wi - (ks - 1) + 2 * pad
    return 2.0 * B * H * W * cout * ks * ks * cin


@_timed("convkxk_mfma", _kxk_flops)
def convkxk(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, ks: int, pad: int, *,
            act: bool = False, slope: float = 0.0, out: Optional[torch.Tensor] = None,
            out2: Optional[torch.Tensor] = None, bias_mod: int = 0) -> torch.Tensor:
    """KS x KS-tap convolution of the 4x4 layers (see s2s_convkxk_nhwc): ks = 2 / pad 0 | 1 = stride-2 convolution on the
    space-to-depth image | transposed form; ks = 4 / pad 1 | 2 = stride-1 convolution | its data gradient.
    ``act``: LeakyReLU(slope) on the bias-added output; ``out2`` receives relu(output) (a view with its own stride)."""
    B, Hi, Wi, cin = x.shape
    H, W = Hi - (ks - 1) + 2 * pad, Wi - (ks - 1) + 2 * pad
    if w_packed.dtype != x.dtype or w_packed.numel() != ((cin + 31) // 32) * ks * ks * cout * 32:
        raise RuntimeError("stain2stain_amd: convkxk packed weight does not match (cout, cin, ks, dtype)")
    px, ldx = _nhwc(x)
    y = torch.empty((B, H, W, cout), dtype=x.dtype, device=x.device) if out is None else out
    if tuple(y.shape) != (B, H, W, cout) or y.dtype != x.dtype:
        raise RuntimeError("stain2stain_amd: convkxk output buffer mismatch")
    py, ldy = _nhwc(y)
    p2, ld2 = 0, 8
    if out2 is not None:
        if tuple(out2.shape) != (B, H, W, cout) or out2.dtype != x.dtype:
            raise RuntimeError("stain2stain_amd: convkxk second output mismatch")
        p2, ld2 = _nhwc(out2)
    if bias is not None and bias.numel() != (bias_mod or cout):
        raise RuntimeError("stain2stain_amd: convkxk bias length does not match (cout, bias_mod)")
    nsplit = _L().s2s_convkxk_ksplit(_dt(x), B, H, W, cout, cin, ks)
    _native.check(min(nsplit, 0), "convkxk_ksplit")
    kwork = torch.empty((nsplit, B * H * W, cout), dtype=torch.float32, device=x.device) if nsplit > 1 else None
    rc = _L().s2s_convkxk_nhwc(_dt(x), px, ldx, cin, _ptr(w_packed), _f32(bias), int(bias_mod), py, ldy, p2, ld2, int(act),
                               float(slope), 0, _f32(kwork), B, H, W, cout, ks, pad, _stream())
    _native.check(rc, "convkxk")
    return y


def _kxk_wgrad_flops(dy, x, grad, ks, accumulate=False, x_plain=False, **kw) -> float:
    B, H, W, cout = dy.shape
    return 2.0 * B * H * W * cout * 16 * (x.shape[3] if (x_plain or ks == 4) else x.shape[3] // 4)


@_timed("convkxk_wgrad_mfma", _kxk_wgrad_flops)
def convkxk_wgrad(dy: torch.Tensor, x: torch.Tensor, grad: torch.Tensor, ks: int, accumulate: bool = False,
                  x_plain: bool = False) -> None:
    """grad (fp32, nn.Conv2d layout [Cout][C][4][4]) (+)= weight gradient: dy [B,H,W,Cout]; x [B,H+1,W+1,cin] with
    cin = 4 C (ks = 2, space-to-depth image) or C (ks = 4); ``x_plain`` (ks = 2, bf16): x is the plain [B,2H,2W,C] tensor
    and the kernel's loader does the space-to-depth."""
    B, H, W, cout = dy.shape
    cin = 4 * x.shape[3] if x_plain else x.shape[3]
    C = cin // 4 if ks == 2 else cin
    if x.dtype != dy.dtype or tuple(x.shape[:3]) != ((B, 2 * H, 2 * W) if x_plain else (B, H + 1, W + 1)):
        raise RuntimeError("stain2stain_amd: convkxk_wgrad operand mismatch")
    if grad.numel() != cout * C * 16 or grad.dtype != torch.float32 or not grad.is_contiguous():
        raise RuntimeError("stain2stain_amd: convkxk_wgrad gradient buffer has the wrong size")
    pdy, lddy = _nhwc(dy)
    px, ldx = _nhwc(x)
    s = _L().s2s_convkxk_wgrad_splits(_dt(dy), B, H, W, cin, cout, ks)
    _native.check(min(s, 0), "convkxk_wgrad_splits")
    part = _workspace(dy.device, s * ks * ks * cout * cin, "wgrad")
    rc = _L().s2s_convkxk_wgrad_nhwc(_dt(dy), pdy, lddy, cout, px, ldx, cin, _f32(part), grad.data_ptr(), 1,
                                     int(accumulate), B, H, W, ks, int(x_plain), _stream())
    _native.check(rc, "convkxk_wgrad")


class SplitSum:
    """The partial slabs of a split-K stride-2 convolution whose reduce launch was left out (``defer=True``): the consumer
    -- the single-launch InstanceNorm of the inner U-Net levels -- folds them itself.  ``slabs``: fp32 [S, B*H*W, C]."""

    def __init__(self, slabs: torch.Tensor, bias: Optional[torch.Tensor], shape, dtype: torch.dtype):
        self.slabs, self.bias, self.shape, self.dtype = slabs, bias, tuple(shape), dtype


def instnorm_split_ok(dtype: torch.dtype, H: int, W: int, C: int) -> bool:
    return dtype == torch.bfloat16 and bool(_L().s2s_instnorm_split_ok(BF16, int(H), int(W), int(C)))


def instnorm_lrelu_fwd2_split(sp: SplitSum, slope: float, out: torch.Tensor, out2: Optional[torch.Tensor] = None,
                              eps: float = 1e-5):
    """instnorm_lrelu_fwd2 on the tensor ``sp`` stands for; returns (stats, raw): raw = that tensor, as the backward needs
    it.  One launch instead of reduce + norm, the same bits."""
    B, H, W, C = sp.shape
    raw = torch.empty((B, H, W, C), dtype=sp.dtype, device=sp.slabs.device)
    py, ldy = _nhwc(out)
    p2, ld2 = (0, 8) if out2 is None else _nhwc(out2)
    stats = torch.empty((4, B, C), dtype=torch.float32, device=raw.device)
    rc = _L().s2s_instnorm_lrelu_fwd_split(_dt(raw), _f32(sp.slabs), sp.slabs.shape[0], _f32(sp.bias), raw.data_ptr(), C,
                                           py, ldy, p2, ld2, _f32(stats), B, H, W, C, eps, slope, _stream())
    _native.check(rc, "instnorm_lrelu_fwd_split")
    return stats, raw


@_timed("instnorm_lrelu_fwd")
def instnorm_lrelu_fwd2(x: torch.Tensor, slope: float, out: torch.Tensor, out2: Optional[torch.Tensor] = None,
                        eps: float = 1e-5):
    """out = leaky_relu(instance_norm(x), slope), out2 (optional view) = relu(instance_norm(x)); returns stats[4][B][C]."""
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    py, ldy = _nhwc(out)
    p2, ld2 = (0, 8) if out2 is None else _nhwc(out2)
    nb = _L().s2s_instnorm_blocks(B, H, W, C)
    _native.check(min(nb, 0), "instnorm_blocks")
    work = torch.empty(2 * B * C * nb, dtype=torch.float32, device=x.device)
    stats = torch.empty((4, B, C), dtype=torch.float32, device=x.device)
    rc = _L().s2s_instnorm_lrelu_fwd2(_dt(x), px, ldx, 0, 0, py, ldy, p2, ld2, _f32(work), _f32(stats), B, H, W, C, eps,
                                      slope, _stream())
    _native.check(rc, "instnorm_lrelu_fwd2")
    return stats


@_timed("instnorm_lrelu_bwd")
def instnorm_lrelu_bwd2(g: torch.Tensor, g2: Optional[torch.Tensor], x: torch.Tensor, stats: torch.Tensor,
                        slope: float) -> torch.Tensor:
    """dx of InstanceNorm + LeakyReLU for the gradients g (wrt the LeakyReLU output; a tensor or the ``SplitSum`` of the
    data gradient that produced it) and g2 (wrt the relu copy)."""
    B, H, W, C = x.shape
    p2, ld2 = (0, 8) if g2 is None else _nhwc(g2)
    px, ldx = _nhwc(x)
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    if isinstance(g, SplitSum):
        if g.shape != (B, H, W, C):
            raise RuntimeError("stain2stain_amd: split gradient does not match the activation")
        rc = _L().s2s_instnorm_lrelu_bwd_split(_dt(x), _f32(g.slabs), g.slabs.shape[0], p2, ld2, px, ldx, _f32(stats),
                                               dx.data_ptr(), C, B, H, W, C, slope, _stream())
        _native.check(rc, "instnorm_lrelu_bwd_split")
        return dx
    pg, ldg = _nhwc(g)
    nb = _L().s2s_instnorm_blocks(B, H, W, C)
    _native.check(min(nb, 0), "instnorm_blocks")
    work = torch.empty(2 * B * C * nb + 2 * B * C, dtype=torch.float32, device=x.device)
    rc = _L().s2s_instnorm_lrelu_bwd2(_dt(x), pg, ldg, p2, ld2, px, ldx, _f32(stats), dx.data_ptr(), C, 0, 0, 0,
                                      _f32(work), B, H, W, C, slope, _stream())
    _native.check(rc, "instnorm_lrelu_bwd2")
    return dx


def pack_conv4x4_t(w: torch.Tensor, stride: int, dtype: torch.dtype):
    """pack_conv4x4 with the operands in ``dtype`` (bf16 or fp32)."""
    cout, cin = w.shape[:2]
    taps, K = (4, 4 * cin) if stride == 2 else (16, cin)
    wf = torch.empty(((K + 31) // 32, taps, cout, 32), dtype=dtype, device=w.device)
    wd = torch.empty(((cout + 31) // 32, taps, K, 32), dtype=dtype, device=w.device)
    rc = _L().s2s_pack_conv4x4_t(_dt(wf), _f32(w.detach().contiguous()), wf.data_ptr(), wd.data_ptr(), cout, cin, stride,
                                 _stream())
    _native.check(rc, "pack_conv4x4_t")
    return wf, wd


@_timed("pack_conv4x4")
def pack_conv4x4_batched(desc: torch.Tensor, total: int, dtype: torch.dtype) -> None:
    """desc: int64 [nlayers, 7] device tensor {w, wf, wd, Cout, Cin, stride == 2, first block}."""
    code = BF16 if dtype == torch.bfloat16 else F32
    if not desc.is_cuda:
        raise RuntimeError("stain2stain_amd: the layer descriptor table must live on the GPU")
    _native.check(_L().s2s_pack_conv4x4_batched(code, desc.data_ptr(), desc.shape[0], int(total), _stream()),
                  "pack_conv4x4_batched")


def space_to_depth_pad1_t(x: torch.Tensor) -> torch.Tensor:
    """space_to_depth_pad1 for either dtype."""
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    xs = torch.empty((B, H // 2 + 1, W // 2 + 1, 4 * C), dtype=x.dtype, device=x.device)
    rc = _L().s2s_space_to_depth_pad1(_dt(x), px, ldx, xs.data_ptr(), 4 * C, 0, B, H, W, C, _stream())
    _native.check(rc, "space_to_depth_pad1")
    return xs


def depth_to_space_unpad1_t(xs: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, Hs, Ws, C4 = xs.shape
    H, W, C = 2 * (Hs - 1), 2 * (Ws - 1), C4 // 4
    pxs, ldxs = _nhwc(xs)
    x = torch.empty((B, H, W, C), dtype=xs.dtype, device=xs.device) if out is None else out
    px, ldx = _nhwc(x)
    rc = _L().s2s_space_to_depth_pad1(_dt(xs), px, ldx, pxs, ldxs, 1, B, H, W, C, _stream())
    _native.check(rc, "depth_to_space_unpad1")
    return x


def p2p_pack_input(a: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
    """out[B,H,W,8] <- [a | b | 0]; a, b NCHW fp32."""
    B, ca, H, W = a.shape
    cb = 0 if b is None else b.shape[1]
    po, ldo = _nhwc(out)
    rc = _L().s2s_p2p_pack_input(_dt(out), _f32(a), ca, _f32(b), cb, po, ldo, B, H, W, _stream())
    _native.check(rc, "p2p_pack_input")
    return out


def p2p_unpack(g: torch.Tensor, c0: int, C: int) -> torch.Tensor:
    """NCHW fp32 [B,C,H,W] <- channels c0 .. c0+C-1 of the 8-channel NHWC image g (an input gradient)."""
    B, H, W, _ = g.shape
    pg, ldg = _nhwc(g)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
    _native.check(_L().s2s_p2p_unpack(_dt(g), pg, ldg, int(c0), int(C), out.data_ptr(), B, H, W, _stream()), "p2p_unpack")
    return out


@_timed("p2p_tanh_l1_fwd")
def p2p_tanh_l1_fwd(h: torch.Tensor, src: torch.Tensor, tgt: torch.Tensor, d_in: torch.Tensor,
                    fake_nchw: Optional[torch.Tensor] = None, l1_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fake = tanh(h[..., :C]); d_in <- [src | fake | 0]; returns (and writes to ``l1_out[0]``) mean |fake - tgt|."""
    B, H, W, _ = h.shape
    C = tgt.shape[1]
    ph, ldh = _nhwc(h)
    pd, ldd = _nhwc(d_in)
    nb = _L().s2s_p2p_tanh_l1_blocks(B, H, W)
    work = torch.empty((nb,), dtype=torch.float64, device=h.device)
    l1 = torch.empty((1,), dtype=torch.float32, device=h.device) if l1_out is None else l1_out
    rc = _L().s2s_p2p_tanh_l1_fwd(_dt(h), ph, ldh, _f32(src), _f32(tgt), pd, ldd, _f32(fake_nchw), _f32(l1),
                                  work.data_ptr(), B, H, W, C, _stream())
    _native.check(rc, "p2p_tanh_l1_fwd")
    return l1


@_timed("p2p_tanh_l1_bwd")
def p2p_tanh_l1_bwd(h: torch.Tensor, tgt: torch.Tensor, gd: Optional[torch.Tensor], l1_scale: float) -> torch.Tensor:
    B, H, W, _ = h.shape
    C = tgt.shape[1]
    ph, ldh = _nhwc(h)
    pg, ldg = (0, 8) if gd is None else _nhwc(gd)
    dh = torch.empty((B, H, W, 8), dtype=h.dtype, device=h.device)
    rc = _L().s2s_p2p_tanh_l1_bwd(_dt(h), ph, ldh, _f32(tgt), pg, ldg, float(l1_scale), dh.data_ptr(), 8, B, H, W, C,
                                  _stream())
    _native.check(rc, "p2p_tanh_l1_bwd")
    return dh


def p2p_bce_logits(z: torch.Tensor, n_real: int, w_real: float, w_fake: float, want_grad: bool = True,
                   out: Optional[torch.Tensor] = None):
    """z [N,H,W,8] (logit = channel 0): (out[2] = mean softplus(-z) over the first n_real samples, mean softplus(z) over
    the others; dz [N,H,W,8] or None)."""
    N, H, W, _ = z.shape
    pz, ldz = _nhwc(z)
    dz = torch.empty((N, H, W, 8), dtype=z.dtype, device=z.device) if want_grad else None
    if out is None:
        out = torch.empty((2,), dtype=torch.float32, device=z.device)
    nb = _L().s2s_p2p_bce_blocks(N, H * W)
    _native.check(min(nb, 0), "p2p_bce_blocks")
    work = _workspace(z.device, 4 * nb, "p2p_bce")            # double[2][nb]
    rc = _L().s2s_p2p_bce_logits_w(_dt(z), pz, ldz, int(n_real), float(w_real), float(w_fake),
                                   0 if dz is None else dz.data_ptr(), 8, _f32(out), work.data_ptr(), N, H * W, _stream())
    _native.check(rc, "p2p_bce_logits")
    return out, dz


@_timed("p2p_act_bwd")
def p2p_act_bwd(g: torch.Tensor, g2: Optional[torch.Tensor], a: torch.Tensor, slope: float,
                dbias: Optional[torch.Tensor]) -> torch.Tensor:
    """dz = a > 0 ? g + g2 : slope * g; dbias (fp32 [C]) <- per-channel sum of dz."""
    B, H, W, C = a.shape
    npix = B * H * W
    pg, ldg = _nhwc(g)
    p2, ld2 = (0, 8) if g2 is None else _nhwc(g2)
    pa, lda = _nhwc(a)
    dz = torch.empty((B, H, W, C), dtype=a.dtype, device=a.device)
    nb = _L().s2s_p2p_act_bwd_blocks(npix, C)
    _native.check(min(nb, 0), "p2p_act_bwd_blocks")
    work = torch.empty((C * nb,), dtype=torch.float32, device=a.device)
    rc = _L().s2s_p2p_act_bwd(_dt(a), pg, ldg, p2, ld2, pa, lda, float(slope), dz.data_ptr(), C, _f32(work),
                              0 if dbias is None else dbias.data_ptr(), 0, npix, C, _stream())
    _native.check(rc, "p2p_act_bwd")
    return dz


def channel_sum_into(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[C] (fp32) <- per-channel sum of an NHWC tensor (conv-bias gradient)."""
    B, H, W, C = x.shape
    px, ldx = _nhwc(x)
    npix = B * H * W
    nb = _L().s2s_channel_sum_blocks(npix, C)
    _native.check(min(nb, 0), "channel_sum_blocks")
    work = torch.empty(2 * C * nb, dtype=torch.float32, device=x.device)
    rc = _L().s2s_channel_sum(_dt(x), px, ldx, _f32(work), out.data_ptr(), npix, C, 0, _stream())
    _native.check(rc, "channel_sum")


def fused_s2_ok(dtype: torch.dtype, c: int) -> bool:
    """The layout-free 4x4 stride-2 kernels (space-to-depth in the loader) take bf16 and a power-of-two channel count."""
    return dtype == torch.bfloat16 and c >= 8 and (c & (c - 1)) == 0


@_timed("convkxk_mfma", lambda x, w_packed, bias, cout, **kw: 2.0 * x.shape[0] * (x.shape[1] // 2) * (x.shape[2] // 2) * cout
        * 16 * x.shape[3])
def conv4x4s2(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, *, act: bool = False,
              slope: float = 0.0, out2: Optional[torch.Tensor] = None, defer: bool = False):
    """nn.Conv2d(k=4, s=2, p=1) from the plain NHWC bf16 input [B,2H,2W,Cin] (no space-to-depth pass).  ``defer``: where the
    launch is split-K and the output is a map the single-launch InstanceNorm takes, return the ``SplitSum`` of its partial
    slabs instead of launching the reduce."""
    B, Hi, Wi, cin = x.shape
    H, W = Hi // 2, Wi // 2
    if Hi % 2 or Wi % 2 or not fused_s2_ok(x.dtype, cin) or w_packed.dtype != x.dtype:
        raise RuntimeError("stain2stain_amd: conv4x4s2 wants bf16, even sizes and a power-of-two channel count")
    if w_packed.numel() != ((4 * cin + 31) // 32) * 4 * cout * 32:
        raise RuntimeError("stain2stain_amd: conv4x4s2 packed weight has the wrong size")
    px, ldx = _nhwc(x)
    y = torch.empty((B, H, W, cout), dtype=x.dtype, device=x.device)
    p2, ld2 = (0, 8) if out2 is None else _nhwc(out2)
    if out2 is not None and tuple(out2.shape) != (B, H, W, cout):
        raise RuntimeError("stain2stain_amd: conv4x4s2 second output mismatch")
    nsplit = _L().s2s_conv4x4s2_ksplit(B, H, W, cout, cin)
    _native.check(min(nsplit, 0), "conv4x4s2_ksplit")
    kwork = torch.empty((nsplit, B * H * W, cout), dtype=torch.float32, device=x.device) if nsplit > 1 else None
    defer = defer and nsplit > 1 and not act and out2 is None and instnorm_split_ok(x.dtype, H, W, cout)
    rc = _L().s2s_conv4x4s2_nhwc(_dt(x), px, ldx, cin, _ptr(w_packed), _f32(bias), 0 if defer else y.data_ptr(), cout, p2,
                                 ld2, int(act), float(slope), _f32(kwork), B, H, W, cout, _stream())
    _native.check(rc, "conv4x4s2")
    return SplitSum(kwork, bias, (B, H, W, cout), x.dtype) if defer else y


@_timed("convkxk_mfma", lambda x, w_packed, bias, cout, **kw: 2.0 * x.shape[0] * 4 * x.shape[1] * x.shape[2] * cout * 4
        * x.shape[3])
def convT4x4s2(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, *,
               out: Optional[torch.Tensor] = None, defer: bool = False):
    """nn.ConvTranspose2d(k=4, s=2, p=1) (= the stride-2 convolution's data gradient) by sub-pixel phase: NHWC bf16
    [B,h,w,Cin] -> plain [B,2h,2w,cout], cout % 64 == 0; w_packed = the data-gradient operand."""
    B, h, w, cin = x.shape
    if x.dtype != torch.bfloat16 or w_packed.dtype != x.dtype or cout % 64:
        raise RuntimeError("stain2stain_amd: convT4x4s2 wants bf16 and cout % 64 == 0")
    if w_packed.numel() != ((cin + 31) // 32) * 4 * 4 * cout * 32:
        raise RuntimeError("stain2stain_amd: convT4x4s2 packed weight has the wrong size")
    px, ldx = _nhwc(x)
    y = torch.empty((B, 2 * h, 2 * w, cout), dtype=x.dtype, device=x.device) if out is None else out
    if tuple(y.shape) != (B, 2 * h, 2 * w, cout) or y.dtype != x.dtype:
        raise RuntimeError("stain2stain_amd: convT4x4s2 output buffer mismatch")
    py, ldy = _nhwc(y)
    nsplit = _L().s2s_convt4x4s2_ksplit(B, h, w, cout, cin)
    _native.check(min(nsplit, 0), "convT4x4s2_ksplit")
    kwork = torch.empty((nsplit, B * 4 * h * w, cout), dtype=torch.float32, device=x.device) if nsplit > 1 else None
    defer = defer and nsplit > 1 and out is None and instnorm_split_ok(x.dtype, 2 * h, 2 * w, cout)      # (see conv4x4s2)
    rc = _L().s2s_convt4x4s2_nhwc(_dt(x), px, ldx, cin, _ptr(w_packed), _f32(bias), 0 if defer else py, ldy, _f32(kwork), B,
                                  h, w, cout, _stream())
    _native.check(rc, "convT4x4s2")
    return SplitSum(kwork, bias, (B, 2 * h, 2 * w, cout), x.dtype) if defer else y


# ------------------------------------------------------------------------------------------------
# the generator's inner levels: convolution + InstanceNorm / activation (or their backward) in one launch (conv_small.hip)
# ------------------------------------------------------------------------------------------------
def convsm_ok(dtype: torch.dtype, mode: int, B: int, h: int, w: int, cin: int, cout: int) -> bool:
    """Does s2s_convsm_nhwc take this layer?  mode 1: h x w = output map of the stride-2 convolution; mode 2: input map of
    the transposed form."""
    return dtype == torch.bfloat16 and bool(_L().s2s_convsm_ok(BF16, int(mode), int(B), int(h), int(w), int(cin), int(cout)))


def _convsm_flops(mode, x, w_packed, bias, cout, **kw) -> float:
    B, hi, wi, cin = x.shape
    return 2.0 * B * (hi // 2) * (wi // 2) * cout * 16 * cin if mode == 1 else 2.0 * B * 4 * hi * wi * cout * 4 * cin


def _convsm_shapes(mode: int, x: torch.Tensor, w_packed: torch.Tensor, cout: int):
    B, hi, wi, cin = x.shape
    if x.dtype != torch.bfloat16 or w_packed.dtype != x.dtype:
        raise RuntimeError("stain2stain_amd: convsm wants bf16 operands")
    if mode == 1:
        if hi % 2 or wi % 2:
            raise RuntimeError("stain2stain_amd: convsm mode 1 wants an even input size")
        h, w, Ho, Wo = hi // 2, wi // 2, hi // 2, wi // 2
    else:
        h, w, Ho, Wo = hi, wi, 2 * hi, 2 * wi
    if w_packed.numel() != 16 * cin * cout:
        raise RuntimeError("stain2stain_amd: convsm packed weight does not match (cin, cout)")
    if not convsm_ok(x.dtype, mode, B, h, w, cin, cout):
        raise RuntimeError("stain2stain_amd: convsm does not take this shape (see s2s_convsm_ok)")
    return B, h, w, Ho, Wo, cin


@_timed("convkxk_mfma", _convsm_flops)
def convsm_fwd(mode: int, x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, *,
               norm: bool, slope: float = 0.0, act: bool = False, out: Optional[torch.Tensor] = None,
               out2: Optional[torch.Tensor] = None, eps: float = 1e-5):
    """One launch for conv (mode 1: Conv2d(4, 2, 1) of the plain x; mode 2: ConvTranspose2d(4, 2, 1)) + bias and
    ``norm``: InstanceNorm + LeakyReLU(slope) -> (y, raw, stats) with out2 (optional view) = relu(norm(raw)); else
    (y, None, None) with y = act ? lrelu(conv + bias, slope) : conv + bias and out2 = relu(y)."""
    B, h, w, Ho, Wo, cin = _convsm_shapes(mode, x, w_packed, cout)
    px, ldx = _nhwc(x)
    y = torch.empty((B, Ho, Wo, cout), dtype=x.dtype, device=x.device) if out is None else out
    if tuple(y.shape) != (B, Ho, Wo, cout) or y.dtype != x.dtype:
        raise RuntimeError("stain2stain_amd: convsm output buffer mismatch")
    py, ldy = _nhwc(y)
    p2, ld2 = 0, 8
    if out2 is not None:
        if tuple(out2.shape) != (B, Ho, Wo, cout) or out2.dtype != x.dtype:
            raise RuntimeError("stain2stain_amd: convsm second output mismatch")
        p2, ld2 = _nhwc(out2)
    raw = stats = None
    if norm:
        raw = torch.empty((B, Ho, Wo, cout), dtype=x.dtype, device=x.device)
        stats = torch.empty((4, B, cout), dtype=torch.float32, device=x.device)
    rc = _L().s2s_convsm_nhwc(BF16, int(mode), px, ldx, cin, _ptr(w_packed), _f32(bias), 1 if norm else 0, int(act),
                              float(slope), float(eps), _ptr(raw), cout, py, ldy, p2, ld2, _f32(stats), 0, 0, 8, 0, 8, 0,
                              B, h, w, cout, _stream())
    _native.check(rc, "convsm_fwd")
    return y, raw, stats


@_timed("convkxk_mfma", lambda mode, g, w_packed, cout, **kw: _convsm_flops(mode, g, w_packed, None, cout))
def convsm_bwd(mode: int, g: torch.Tensor, w_packed: torch.Tensor, cout: int, *, z: Optional[torch.Tensor],
               stats: Optional[torch.Tensor], g2: Optional[torch.Tensor], slope: float, bwd_c0: int = 0):
    """Data gradient (mode 2 with wd: of a stride-2 convolution; mode 1 with wf: of a transposed one) of ``cout`` channels
    ending in the InstanceNorm + activation backward of the tensor those channels belong to: channels [bwd_c0, cout) are
    the gradient wrt lrelu(norm(z), slope) (plus ``g2`` wrt relu(norm(z))) and come back as dz; channels [0, bwd_c0) come
    back as they are (the skip half of a decoder input).  Returns (dz or None, plain or None)."""
    B, h, w, Ho, Wo, cin = _convsm_shapes(mode, g, w_packed, cout)
    pg, ldg = _nhwc(g)
    C2 = cout - bwd_c0
    dz = plain = None
    pz = ldz = ps = p2 = 0
    ldz = ld2 = 8
    pdz, lddz, ppl, ldpl = 0, 8, 0, 8
    if C2 > 0:
        if z is None or stats is None or tuple(z.shape) != (B, Ho, Wo, C2) or tuple(stats.shape) != (4, B, C2):
            raise RuntimeError("stain2stain_amd: convsm_bwd wants z [B,H,W,C] and stats [4,B,C] of the normed tensor")
        pz, ldz = _nhwc(z)
        ps = _f32(stats)
        if g2 is not None:
            if tuple(g2.shape) != (B, Ho, Wo, C2) or g2.dtype != g.dtype:
                raise RuntimeError("stain2stain_amd: convsm_bwd second gradient mismatch")
            p2, ld2 = _nhwc(g2)
        dz = torch.empty((B, Ho, Wo, C2), dtype=g.dtype, device=g.device)
        pdz, lddz = _nhwc(dz)
    if bwd_c0 > 0:
        plain = torch.empty((B, Ho, Wo, bwd_c0), dtype=g.dtype, device=g.device)
        ppl, ldpl = _nhwc(plain)
    rc = _L().s2s_convsm_nhwc(BF16, int(mode), pg, ldg, cin, _ptr(w_packed), 0, 2, 0, float(slope), 0.0, 0, 8, pdz, lddz,
                              ppl, ldpl, 0, ps, pz, ldz, p2, ld2, int(bwd_c0), B, h, w, cout, _stream())
    _native.check(rc, "convsm_bwd")
    return dz, plain


def convsm_wins(dtype: torch.dtype, mode: int, B: int, h: int, w: int, cin: int, cout: int) -> bool:
    """Launch policy of the inner levels: where the sample-complete launch measured faster than split-K convolution +
    single-launch InstanceNorm at batch 16 (scripts/p2p_small_bench.py, rocprofv3 kernel times).  With the samples' input
    staged in LDS (s2s_convsm_ok() == 2) it takes every layer it fits; the form that streams the im2col'd pixels from L2
    only wins on the smallest maps."""
    if dtype != torch.bfloat16:
        return False
    how = int(_L().s2s_convsm_ok(BF16, int(mode), int(B), int(h), int(w), int(cin), int(cout)))
    if how == 2:
        return True
    if how == 0:
        return False
    if mode == 1:
        return h * w <= 4 or (h * w <= 16 and cout <= 512)
    return h * w <= 4
