"""Python mirror of the batched NS C-ABI (include/asp_ns.h) over ctypes.

This is plumbing only: every call goes straight into libasp_amd.so (the
hand-written HIP path).  There is no CPU fallback -- if the library is missing
or no HIP device is present the constructor raises.

Method names mirror the reference's verbs (WebRtcNs_Init / _set_policy /
_Analyze / _Process, noise_suppression.h:16-122) with a leading stream count.
"""
import ctypes as C
import os

import numpy as np

from ._abi import BLOCKL, MEM_DEVICE, MEM_HOST, AspNsState
from .build import LIB

_lib = None


class AspError(RuntimeError):
    pass


def load_library():
    """dlopen lib/libasp_amd.so (built by build.py / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        # a fresh checkout: compile the library in-tree (hipcc); there is no CPU fallback
        try:
            from .build import build_library
            build_library()
        except Exception as e:  # noqa: BLE001
            raise AspError(
                "HIP library %s is missing and could not be built (%s): run `python -c 'import "
                "__graft_entry__ as g; g.build()'` (there is no CPU fallback)" % (LIB, e))
    lib = C.CDLL(LIB)
    vp, ip, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    sig = {
        "AspNsBatch_Create": [C.POINTER(vp), ip, ip],
        "AspNsBatch_Free": [vp],
        "AspNsBatch_Init": [vp, C.c_uint32],
        "AspNsBatch_set_policy": [vp, ip],
        "AspNsBatch_InitStream": [vp, ip],
        "AspNsBatch_set_policy_stream": [vp, ip, ip],
        "AspNsBatch_num_streams": [vp],
        "AspNsBatch_Analyze": [vp, vp, ip],
        "AspNsBatch_Process": [vp, vp, vp, ip],
        "AspNsBatch_AnalyzeProcess": [vp, vp, vp, ip, ip],
        "AspNsBatch_AnalyzeProcessS16": [vp, vp, vp, ip, ip],
        "AspNsBatch_AnalyzeProcessBands": [vp, vp, vp, vp, vp, ip, ip],
        "AspNsBatch_ProcessBands": [vp, vp, vp, vp, vp, ip],
        "AspNsBatch_num_bands": [vp],
        "AspNsBatch_ExportHbState": [vp, ip, vp],
        "AspNsBatch_ImportHbState": [vp, ip, vp],
        "AspNsBatch_ExportState": [vp, ip, C.POINTER(AspNsState)],
        "AspNsBatch_ImportState": [vp, ip, C.POINTER(AspNsState)],
        "AspNsBatch_prior_speech_probability": [vp, vp],
        "AspNsBatch_SetStream": [vp, vp],
        "AspNsBatch_SetSplit": [vp, ip],
        "AspNsBatch_SetFlow": [vp, ip],
        "AspNsBatch_DebugFlowDesync": [vp],
        "AspNsBatch_SetKernel": [vp, ip],
        "AspNsBatch_Synchronize": [vp],
        "AspNsBatch_TimedSteps": [vp, vp, vp, ip, ip, fp],
        "AspNsBatch_LastEnqueueUs": [vp, C.POINTER(C.c_double)],
        "AspNsBatch_AnalyzeProcessReplay": [vp, vp, vp, ip, ip],
        "AspNsBatch_SetGraph": [vp, ip],
        "AspNs_CopyCeiling": [C.c_size_t, ip, ip, C.POINTER(C.c_double)],
        "AspNs_DeviceAlloc": [C.POINTER(vp), C.c_size_t, ip],
        "AspNs_DeviceFree": [vp],
        "AspNs_MemcpyH2D": [vp, vp, C.c_size_t],
        "AspNs_MemcpyD2H": [vp, vp, C.c_size_t],
        "AspNs_rdft256_batch": [vp, ip, ip, ip, ip],
        "AspNs_rdft128_batch": [vp, ip, ip, ip, ip],
        "AspNs_device_count": [],
    }
    for name, args in sig.items():
        if os.environ.get("ASP_AMD_LIB") and not hasattr(lib, name):
            continue  # an earlier build of the library loaded for a same-box A / B run (build.py)
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.AspNsBatch_GetStream.argtypes = [vp]
    lib.AspNsBatch_GetStream.restype = vp
    lib.AspNs_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise AspError("%s failed (%d): %s" % (what, rc, load_library().AspNs_last_error().decode()))


def copy_ceiling_gbs(nbytes=1 << 30, iters=8, device=0):
    """Measured streaming-copy rate of the box in GB/s (read + written bytes, float4 copy kernel)."""
    g = C.c_double()
    _check(load_library().AspNs_CopyCeiling(nbytes, iters, device, C.byref(g)), "AspNs_CopyCeiling")
    return g.value


def device_count():
    n = load_library().AspNs_device_count()
    return max(n, 0)


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class DeviceBuffer:
    """Raw device allocation owned by the library (for callers without torch)."""

    def __init__(self, nbytes, device=0):
        self.lib = load_library()
        p = C.c_void_p()
        _check(self.lib.AspNs_DeviceAlloc(C.byref(p), nbytes, device), "AspNs_DeviceAlloc")
        self.ptr = p.value
        self.nbytes = nbytes

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(self.lib.AspNs_MemcpyH2D(self.ptr, _ptr(arr), arr.nbytes), "AspNs_MemcpyH2D")

    def download(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        _check(self.lib.AspNs_MemcpyD2H(_ptr(out), self.ptr, out.nbytes), "AspNs_MemcpyD2H")
        return out

    def free(self):
        if self.ptr:
            self.lib.AspNs_DeviceFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class NsBatch:
    """N independent noise-suppressor streams on one GPU (fs = 8000: 80-sample frames; 16000 and up:
    160-sample frames of the 0-8 kHz band)."""

    def __init__(self, num_streams, device=0, fs=16000, policy=None, kernel=None):
        self.lib = load_library()
        self.S = int(num_streams)
        self.device = device
        h = C.c_void_p()
        _check(self.lib.AspNsBatch_Create(C.byref(h), self.S, device), "AspNsBatch_Create")
        self.h = h
        self.init(fs)
        if policy is not None:
            self.set_policy(policy)
        if kernel is not None:
            self.set_kernel(kernel)

    def init(self, fs=16000):
        _check(self.lib.AspNsBatch_Init(self.h, fs), "AspNsBatch_Init")
        self.block = 80 if fs == 8000 else BLOCKL   # ns_core.c:89-98

    def set_policy(self, mode):
        _check(self.lib.AspNsBatch_set_policy(self.h, mode), "AspNsBatch_set_policy")

    def init_stream(self, stream):
        """WebRtcNs_Init of one stream of the running batch (policy back to 0)."""
        _check(self.lib.AspNsBatch_InitStream(self.h, stream), "AspNsBatch_InitStream")

    def set_policy_stream(self, stream, mode):
        _check(self.lib.AspNsBatch_set_policy_stream(self.h, stream, mode), "AspNsBatch_set_policy_stream")

    # -- host-array convenience (copies in/out, synchronous)
    def analyze(self, frames):
        frames = np.ascontiguousarray(frames, np.float32)
        assert frames.shape == (self.S, self.block)
        _check(self.lib.AspNsBatch_Analyze(self.h, _ptr(frames), MEM_HOST), "AspNsBatch_Analyze")

    def process(self, frames):
        frames = np.ascontiguousarray(frames, np.float32)
        assert frames.shape == (self.S, self.block)
        out = np.empty_like(frames)
        _check(self.lib.AspNsBatch_Process(self.h, _ptr(frames), _ptr(out), MEM_HOST),
               "AspNsBatch_Process")
        return out

    def analyze_process(self, frames):
        """frames [F][S][160] -> denoised [F][S][160] (fused step per frame)."""
        frames = np.ascontiguousarray(frames, np.float32)
        F = frames.shape[0]
        assert frames.shape == (F, self.S, self.block)
        out = np.empty_like(frames)
        _check(self.lib.AspNsBatch_AnalyzeProcess(self.h, _ptr(frames), _ptr(out), F, MEM_HOST),
               "AspNsBatch_AnalyzeProcess")
        return out

    def analyze_process_bands(self, low, high):
        """low [F][S][160], high [F][nh][S][160] float32 -> (out_low, out_high): the fused step of the
        low band plus the high-band gain (32 / 48 kHz batches)."""
        low = np.ascontiguousarray(low, np.float32)
        high = np.ascontiguousarray(high, np.float32)
        F = low.shape[0]
        assert low.shape == (F, self.S, BLOCKL) and high.shape[0] == F and high.shape[2:] == (self.S, BLOCKL)
        ol, oh = np.empty_like(low), np.empty_like(high)
        _check(self.lib.AspNsBatch_AnalyzeProcessBands(self.h, _ptr(low), _ptr(high), _ptr(ol), _ptr(oh), F,
                                                       MEM_HOST), "AspNsBatch_AnalyzeProcessBands")
        return ol, oh

    def process_bands(self, low, high):
        """One frame: WebRtcNs_Process with bands after a separate analyze(); low [S][160],
        high [nh][S][160]."""
        low = np.ascontiguousarray(low, np.float32)
        high = np.ascontiguousarray(high, np.float32)
        ol, oh = np.empty_like(low), np.empty_like(high)
        _check(self.lib.AspNsBatch_ProcessBands(self.h, _ptr(low), _ptr(high), _ptr(ol), _ptr(oh), MEM_HOST),
               "AspNsBatch_ProcessBands")
        return ol, oh

    def export_hb(self, stream):
        from ._abi import AspNsHbState
        hb = AspNsHbState()
        _check(self.lib.AspNsBatch_ExportHbState(self.h, stream, C.byref(hb)), "AspNsBatch_ExportHbState")
        return np.ctypeslib.as_array(hb.dataBufHB).reshape(2, 256).copy()

    def analyze_process_s16(self, pcm):
        """pcm [F][S][160] int16 -> denoised int16 of the same shape (fused step, PCM in/out)."""
        pcm = np.ascontiguousarray(pcm, np.int16)
        F = pcm.shape[0]
        assert pcm.shape == (F, self.S, self.block)
        out = np.empty_like(pcm)
        _check(self.lib.AspNsBatch_AnalyzeProcessS16(self.h, _ptr(pcm), _ptr(out), F, MEM_HOST),
               "AspNsBatch_AnalyzeProcessS16")
        return out

    # -- device-pointer path (asynchronous on the batch's stream)
    def analyze_process_device(self, in_ptr, out_ptr, num_frames):
        _check(self.lib.AspNsBatch_AnalyzeProcess(self.h, C.c_void_p(in_ptr), C.c_void_p(out_ptr),
                                                  num_frames, MEM_DEVICE),
               "AspNsBatch_AnalyzeProcess")

    def analyze_process_replay(self, in_ptr, out_ptr, frames_in_ring, steps):
        """`steps` fused frame steps over a device ring as one replay of a captured hipGraph (async)."""
        _check(self.lib.AspNsBatch_AnalyzeProcessReplay(self.h, C.c_void_p(in_ptr), C.c_void_p(out_ptr),
                                                        frames_in_ring, steps),
               "AspNsBatch_AnalyzeProcessReplay")

    def set_graph(self, on):
        """on: replay captured hipGraphs instead of plain launches."""
        _check(self.lib.AspNsBatch_SetGraph(self.h, 1 if on else 0), "AspNsBatch_SetGraph")

    def timed_steps(self, in_ptr, out_ptr, frames_in_ring, steps):
        """K fused frame steps bracketed by hipEvents on the launch stream -> ms."""
        ms = C.c_float()
        _check(self.lib.AspNsBatch_TimedSteps(self.h, C.c_void_p(in_ptr), C.c_void_p(out_ptr),
                                              frames_in_ring, steps, C.byref(ms)),
               "AspNsBatch_TimedSteps")
        return ms.value

    def last_enqueue_us(self):
        """Host microseconds the last timed_steps() call spent enqueuing its launches."""
        us = C.c_double()
        _check(self.lib.AspNsBatch_LastEnqueueUs(self.h, C.byref(us)), "AspNsBatch_LastEnqueueUs")
        return us.value

    def set_stream(self, hip_stream):
        _check(self.lib.AspNsBatch_SetStream(self.h, C.c_void_p(hip_stream)), "AspNsBatch_SetStream")

    def set_flow(self, mode):
        """Hand-off build of the multi-frame entry points: -1 default, 0 off, 1 on (include/asp_ns.h)."""
        _check(self.lib.AspNsBatch_SetFlow(self.h, mode), "AspNsBatch_SetFlow")

    def set_split(self, parts):
        _check(self.lib.AspNsBatch_SetSplit(self.h, parts), "AspNsBatch_SetSplit")

    def set_kernel(self, kernel):
        """0 / 3: the pair-layout fused kernel (default); 1: the bins q / q + 64 kernel."""
        _check(self.lib.AspNsBatch_SetKernel(self.h, kernel), "AspNsBatch_SetKernel")

    def synchronize(self):
        _check(self.lib.AspNsBatch_Synchronize(self.h), "AspNsBatch_Synchronize")

    def export_state(self, stream):
        s = AspNsState()
        _check(self.lib.AspNsBatch_ExportState(self.h, stream, C.byref(s)), "AspNsBatch_ExportState")
        return s

    def import_state(self, stream, state):
        _check(self.lib.AspNsBatch_ImportState(self.h, stream, C.byref(state)),
               "AspNsBatch_ImportState")

    def prior_speech_probability(self):
        out = np.empty(self.S, np.float32)
        _check(self.lib.AspNsBatch_prior_speech_probability(self.h, _ptr(out)),
               "AspNsBatch_prior_speech_probability")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.AspNsBatch_Free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rdft128(rows, isgn, device=0):
    """WebRtc_rdft(128, isgn), the 8 kHz transform, on every row of `rows` ([count][128] float32)."""
    lib = load_library()
    rows = np.ascontiguousarray(rows, np.float32).copy()
    flat = rows.reshape(-1, 128)
    _check(lib.AspNs_rdft128_batch(_ptr(flat), flat.shape[0], isgn, MEM_HOST, device),
           "AspNs_rdft128_batch")
    return rows


def rdft256(rows, isgn, device=0):
    """WebRtc_rdft(256, isgn) on every row of `rows` ([count][256] float32)."""
    lib = load_library()
    rows = np.ascontiguousarray(rows, np.float32).copy()
    flat = rows.reshape(-1, 256)
    _check(lib.AspNs_rdft256_batch(_ptr(flat), flat.shape[0], isgn, MEM_HOST, device),
           "AspNs_rdft256_batch")
    return rows
