"""Python mirror of the batched echo-canceller C-ABI (include/asp_aec.h) over ctypes.
Plumbing only -- every call goes into libasp_amd.so; no CPU fallback."""
import ctypes as C
import os

import numpy as np

from ._abi import MEM_DEVICE, MEM_HOST, AecConfig, AspAecControl, AspAecState
from .ns import AspError, load_library

_sig_done = False


def _lib():
    global _sig_done
    lib = load_library()
    if not _sig_done:
        vp, ip = C.c_void_p, C.c_int
        sig = {
            "AspAecBatch_Create": [C.POINTER(vp), ip, ip],
            "AspAecBatch_Free": [vp],
            "AspAecBatch_Init": [vp, C.c_int32, C.c_int32],
            "AspAecBatch_set_config": [vp, AecConfig],
            "AspAecBatch_num_streams": [vp],
            "AspAecBatch_enable_delay_correction": [vp, ip],
            "AspAecBatch_enable_reported_delay": [vp, ip],
            "AspAecBatch_reported_delay_enabled": [vp],
            "AspAecBatch_GetDelayMetrics": [vp, vp, vp],
            "AspAecBatch_ExportDelayState": [vp, ip, vp],
            "AspAecBatch_delay_correction_enabled": [vp],
            "AspAecBatch_BufferFarend": [vp, vp, ip, ip],
            "AspAecBatch_Process": [vp, vp, vp, ip, ip, C.c_int32, ip],
            "AspAecBatch_ProcessBands": [vp, vp, vp, vp, vp, ip, ip, C.c_int32, ip],
            "AspAecBatch_num_bands": [vp],
            "AspAecBatch_Run": [vp, vp, vp, vp, ip, ip, ip, ip],
            "AspAecBatch_get_echo_status": [vp, C.POINTER(ip)],
            "AspAecBatch_get_error_code": [vp],
            "AspAecBatch_GetMetrics": [vp, vp],
            "AspAecBatch_ExportMetricsState": [vp, ip, vp],
            "AspAecBatch_ExportState": [vp, ip, C.POINTER(AspAecState)],
            "AspAecBatch_ImportState": [vp, ip, C.POINTER(AspAecState)],
            "AspAecBatch_GetControl": [vp, C.POINTER(AspAecControl)],
            "AspAecBatch_Synchronize": [vp],
            "AspAecBatch_SetFlow": [vp, ip],
            "AspAecBatch_ProcessV": [vp, vp, vp, ip, vp, vp, vp, ip],
            "AspAecBatch_InitStream": [vp, ip],
            "AspAecBatch_GetControlStream": [vp, ip, C.POINTER(AspAecControl)],
            "AspAecBatch_TimedSteps": [vp, vp, vp, vp, ip, ip, ip, C.POINTER(C.c_float)],
            "AspAec_rdft128_batch": [vp, vp, ip, ip, ip],
            "AspAec_host_table": [ip, vp, ip],
            "AspAec_delay_estimator_batch": [vp, ip, vp, vp, ip, ip],
        }
        for name, args in sig.items():
            if os.environ.get("ASP_AMD_LIB") and not hasattr(lib, name):
                continue  # an earlier build of the library loaded for a same-box A / B run (build.py)
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _sig_done = True
    return lib


def _check(rc, what):
    if rc != 0:
        raise AspError("%s failed (%d)" % (what, rc))


class AecBatch:
    """N independent echo-canceller streams, fed in lock-step, on one GPU
    (WebRtcAec_Create + _Init of the reference for every stream)."""

    def __init__(self, num_streams, fs=16000, sc_fs=48000, nlp_mode=None, device=0):
        self.lib = _lib()
        self.S = int(num_streams)
        h = C.c_void_p()
        _check(self.lib.AspAecBatch_Create(C.byref(h), self.S, device), "AspAecBatch_Create")
        self.h = h
        self.init_rc = self.lib.AspAecBatch_Init(self.h, fs, sc_fs)
        if self.init_rc == 0 and nlp_mode is not None:
            assert self.set_config(nlp_mode) == 0

    def set_config(self, nlp_mode, skew=0, metrics=0, delay_logging=0):
        return self.lib.AspAecBatch_set_config(self.h, AecConfig(nlp_mode, skew, metrics, delay_logging))

    def enable_delay_correction(self, enable=1):
        """WebRtcAec_enable_delay_correction for every stream (aec_core.c:1876-1881): the 32-partition extended
        filter and the ProcessExtended delay handling.  Call after construction (Init switches it off)."""
        _check(self.lib.AspAecBatch_enable_delay_correction(self.h, int(enable)), "AspAecBatch_enable_delay_correction")

    def delay_correction_enabled(self):
        return int(self.lib.AspAecBatch_delay_correction_enabled(self.h))

    def enable_reported_delay(self, enable=1):
        """WebRtcAec_enable_reported_delay for every stream; 0 = the delay-agnostic mode."""
        _check(self.lib.AspAecBatch_enable_reported_delay(self.h, int(enable)), "AspAecBatch_enable_reported_delay")

    def delay_metrics(self):
        """(rc, median[S], std[S]) of WebRtcAec_GetDelayMetrics for every stream"""
        med, std = np.full(self.S, -99, np.int32), np.full(self.S, -99, np.int32)
        rc = self.lib.AspAecBatch_GetDelayMetrics(self.h, med.ctypes.data, std.ctypes.data)
        if rc not in (0, -1):
            raise AspError("AspAecBatch_GetDelayMetrics failed (%d)" % rc)
        return rc, med, std

    def delay_state(self, stream):
        from ._abi import AspAecDelayState
        d = AspAecDelayState()
        _check(self.lib.AspAecBatch_ExportDelayState(self.h, stream, C.byref(d)), "AspAecBatch_ExportDelayState")
        return d

    def error_code(self):
        return self.lib.AspAecBatch_get_error_code(self.h)

    def buffer_farend(self, far):
        """far [S][n] float32 (n = 80 / 160); returns the reference's 0 / -1."""
        far = np.ascontiguousarray(far, np.float32)
        return self.lib.AspAecBatch_BufferFarend(self.h, far.ctypes.data, far.shape[-1], MEM_HOST)

    def process(self, near, delay_ms=0, skew=0):
        """near [S][n] -> (out [S][n], rc)."""
        near = np.ascontiguousarray(near, np.float32)
        out = np.empty_like(near)
        rc = self.lib.AspAecBatch_Process(self.h, near.ctypes.data, out.ctypes.data, near.shape[-1],
                                          delay_ms, skew, MEM_HOST)
        return out, rc

    def process_bands(self, near_low, near_high, delay_ms=0, skew=0):
        """32 kHz: near_low / near_high [S][n] -> (out_low, out_high, rc)."""
        nl, nh = np.ascontiguousarray(near_low, np.float32), np.ascontiguousarray(near_high, np.float32)
        ol, oh = np.empty_like(nl), np.empty_like(nh)
        rc = self.lib.AspAecBatch_ProcessBands(self.h, nl.ctypes.data, nh.ctypes.data, ol.ctypes.data,
                                               oh.ctypes.data, nl.shape[-1], delay_ms, skew, MEM_HOST)
        return ol, oh, rc

    def frame_bands(self, far, near_low, near_high, delay_ms=0):
        rc = self.buffer_farend(far)
        ol, oh, rc2 = self.process_bands(near_low, near_high, delay_ms)
        return ol, oh, rc | rc2

    def process_v(self, near, delays_ms):
        """near [S][n], delays_ms [S] (each stream's own reported delay) -> (out [S][n], status [S] int32)."""
        near = np.ascontiguousarray(near, np.float32)
        ms = np.ascontiguousarray(delays_ms, np.int16)
        assert ms.shape == (self.S,)
        out = np.empty_like(near)
        status = np.zeros(self.S, np.int32)
        rc = self.lib.AspAecBatch_ProcessV(self.h, near.ctypes.data, out.ctypes.data, near.shape[-1], ms.ctypes.data,
                                           None, status.ctypes.data, MEM_HOST)
        if rc not in (0, -1):
            raise AspError("AspAecBatch_ProcessV failed (%d)" % rc)
        return out, status

    def init_stream(self, stream):
        """WebRtcAec_Init of one stream of the running batch."""
        _check(self.lib.AspAecBatch_InitStream(self.h, stream), "AspAecBatch_InitStream")

    def control_stream(self, stream):
        c = AspAecControl()
        _check(self.lib.AspAecBatch_GetControlStream(self.h, stream, C.byref(c)), "AspAecBatch_GetControlStream")
        return c

    def frame(self, far, near, delay_ms=0, skew=0):
        rc = self.buffer_farend(far)
        out, rc2 = self.process(near, delay_ms, skew)
        return out, rc | rc2

    def run(self, far, near, delay_ms=0):
        """far / near [F][S][n] -> out [F][S][n]: F x (BufferFarend + Process) in one call."""
        far = np.ascontiguousarray(far, np.float32)
        near = np.ascontiguousarray(near, np.float32)
        assert far.shape == near.shape and far.shape[1] == self.S
        out = np.empty_like(near)
        rc = self.lib.AspAecBatch_Run(self.h, far.ctypes.data, near.ctypes.data, out.ctypes.data,
                                      far.shape[2], far.shape[0], delay_ms, MEM_HOST)
        if rc not in (0, -1):
            raise AspError("AspAecBatch_Run failed (%d)" % rc)
        return out

    def run_device(self, far_ptr, near_ptr, out_ptr, n, num_frames, delay_ms=0):
        rc = self.lib.AspAecBatch_Run(self.h, C.c_void_p(far_ptr), C.c_void_p(near_ptr), C.c_void_p(out_ptr),
                                      n, num_frames, delay_ms, MEM_DEVICE)
        if rc not in (0, -1):
            raise AspError("AspAecBatch_Run failed (%d)" % rc)

    def timed_steps(self, far_ptr, near_ptr, out_ptr, n, frames_in_ring, steps):
        ms = C.c_float()
        _check(self.lib.AspAecBatch_TimedSteps(self.h, C.c_void_p(far_ptr), C.c_void_p(near_ptr),
                                               C.c_void_p(out_ptr), n, frames_in_ring, steps, C.byref(ms)),
               "AspAecBatch_TimedSteps")
        return ms.value

    def synchronize(self):
        _check(self.lib.AspAecBatch_Synchronize(self.h), "AspAecBatch_Synchronize")

    def set_flow(self, mode):
        """Hand-off build of run() / timed_steps(): -1 default, 0 off, 1 on (include/asp_aec.h)."""
        _check(self.lib.AspAecBatch_SetFlow(self.h, mode), "AspAecBatch_SetFlow")

    def export_state(self, stream):
        st = AspAecState()
        _check(self.lib.AspAecBatch_ExportState(self.h, stream, C.byref(st)), "AspAecBatch_ExportState")
        return st

    def metrics_state(self, stream):
        from ._abi import AspAecMetricsState
        m = AspAecMetricsState()
        _check(self.lib.AspAecBatch_ExportMetricsState(self.h, stream, C.byref(m)), "AspAecBatch_ExportMetricsState")
        return m

    def get_metrics(self):
        """WebRtcAec_GetMetrics of every stream: int32 [S][16] = rerl, erl, erle, aNlp x (instant, average, max, min)."""
        from ._abi import AecMetrics
        arr = (AecMetrics * self.S)()
        rc = self.lib.AspAecBatch_GetMetrics(self.h, arr)
        if rc != 0:
            raise AspError("AspAecBatch_GetMetrics failed (%d)" % rc)
        return np.array([m.to_tuple() for m in arr], np.int32)

    def import_state(self, stream, st):
        _check(self.lib.AspAecBatch_ImportState(self.h, stream, C.byref(st)), "AspAecBatch_ImportState")

    def control(self):
        c = AspAecControl()
        _check(self.lib.AspAecBatch_GetControl(self.h, C.byref(c)), "AspAecBatch_GetControl")
        return c

    def echo_status(self):
        st = (C.c_int * self.S)()
        _check(self.lib.AspAecBatch_get_echo_status(self.h, st), "AspAecBatch_get_echo_status")
        return np.array(list(st))

    def close(self):
        if getattr(self, "h", None):
            self.lib.AspAecBatch_Free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def rdft128(rows, isgn, device=0):
    """aec_rdft_forward_128 / inverse_128 on [count][128] float32 rows, on the GPU."""
    lib = _lib()
    rows = np.ascontiguousarray(rows, np.float32)
    out = np.empty_like(rows)
    _check(lib.AspAec_rdft128_batch(rows.ctypes.data, out.ctypes.data, isgn, rows.size // 128, device),
           "AspAec_rdft128_batch")
    return out


def delay_estimator_batch(states, binary_far, binary_near, device=0):
    """`len(states)` binary delay estimators (AspAecDelayState, updated in place), each over its own row of
    binary far / near spectra ([count][nblocks] uint32), on the GPU (include/asp_aec.h)."""
    from ._abi import AspAecDelayState
    lib = _lib()
    far = np.ascontiguousarray(binary_far, np.uint32)
    near = np.ascontiguousarray(binary_near, np.uint32)
    assert far.shape == near.shape and far.shape[0] == len(states)
    arr = (AspAecDelayState * len(states))(*states)
    _check(lib.AspAec_delay_estimator_batch(C.byref(arr), len(states), far.ctypes.data, near.ctypes.data, far.shape[1],
                                            device), "AspAec_delay_estimator_batch")
    for i in range(len(states)):
        C.memmove(C.byref(states[i]), C.byref(arr[i]), C.sizeof(AspAecDelayState))


def host_table(which, n):
    lib = _lib()
    buf = np.zeros(n, np.float32)
    got = lib.AspAec_host_table(which, buf.ctypes.data, n)
    if got != n:
        raise AspError("AspAec_host_table(%d) returned %d" % (which, got))
    return buf
