"""Loaders for the parity checkers under oracle/ (tests and bench baseline only).

OracleNs  -- this repo's CPU restatement (oracle/libasp_oracle.so)
RefNs     -- the reference C compiled in place from /root/reference
             (oracle/_ref/libns_ref.so); absent when it has not been built.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from audiosignalprocess_amd._abi import BLOCKL, AspNsState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libasp_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libns_ref.so")

REDUCE_SEQ = 0
REDUCE_TREE = 1
REDUCE_TREE32 = 2
REDUCE_TREE64P = 3   # ns_kernels1.hip: one stream per wave, pair layout

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build_oracle():
    """(Re)build the checkers; the reference part only where it is present."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _load_oracle():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    lib = C.CDLL(ORACLE_SO)
    sp = C.POINTER(AspNsState)
    lib.asp_ns_oracle_init.argtypes = [sp, C.c_uint32]
    lib.asp_ns_oracle_init.restype = C.c_int
    lib.asp_ns_oracle_set_policy.argtypes = [sp, C.c_int]
    lib.asp_ns_oracle_set_policy.restype = C.c_int
    lib.asp_ns_oracle_run.argtypes = [sp, C.c_int, _f32p, _f32p, C.c_int, C.c_int]
    lib.asp_ns_oracle_run_mt.argtypes = [sp, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
    lib.asp_ns_oracle_analyze.argtypes = [sp, _f32p, C.c_int]
    lib.asp_ns_oracle_process.argtypes = [sp, _f32p, _f32p, C.c_int]
    lib.asp_ns_oracle_rdft256.argtypes = [_f32p, C.c_int]
    lib.asp_ns_oracle_rdft128.argtypes = [_f32p, C.c_int]
    for n in ("window", "fft_w", "fft_c", "window8", "fft_w8", "fft_c8"):
        getattr(lib, "asp_ns_oracle_" + n).restype = C.POINTER(C.c_float)
    return lib


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        _oracle = _load_oracle()
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ref_ns_sizeof.restype = C.c_size_t
        lib.WebRtcNs_InitCore.argtypes = [C.c_void_p, C.c_uint32]
        lib.WebRtcNs_set_policy_core.argtypes = [C.c_void_p, C.c_int]
        lib.WebRtcNs_AnalyzeCore.argtypes = [C.c_void_p, _f32p]
        lib.ref_ns_export.argtypes = [C.c_void_p, C.POINTER(AspNsState)]
        lib.ref_ns_import.argtypes = [C.c_void_p, C.POINTER(AspNsState)]
        lib.ref_ns_run.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.c_int, C.c_int]
        lib.ref_rdft256.argtypes = [_f32p, C.c_int]
        lib.ref_rdft128.argtypes = [_f32p, C.c_int]
        lib.ref_ns_fft_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _ref = lib
    return _ref


def oracle_table(name, n):
    p = getattr(oracle_lib(), "asp_ns_oracle_" + name)()
    return np.ctypeslib.as_array(p, shape=(n,)).copy()


class OracleNs:
    """Batch of streams driven through this repo's CPU restatement."""

    def __init__(self, num_streams, policy=1, reduce_mode=REDUCE_SEQ, fs=16000):
        self.lib = oracle_lib()
        self.S = num_streams
        self.mode = reduce_mode
        self.block = 80 if fs == 8000 else BLOCKL   # ns_core.c:89-98
        self.states = (AspNsState * num_streams)()
        for i in range(num_streams):
            assert self.lib.asp_ns_oracle_init(C.byref(self.states[i]), fs) == 0
            assert self.lib.asp_ns_oracle_set_policy(C.byref(self.states[i]), policy) == 0

    def set_policy(self, policy, stream=None):
        for i in (range(self.S) if stream is None else [stream]):
            assert self.lib.asp_ns_oracle_set_policy(C.byref(self.states[i]), policy) == 0

    def run(self, frames, threads=1):
        """frames [F][S][160] float32 ([F][S][80] at 8 kHz) -> output of the same shape."""
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        F = frames.shape[0]
        assert frames.shape == (F, self.S, self.block)
        out = np.empty_like(frames)
        if threads > 1:
            self.lib.asp_ns_oracle_run_mt(self.states, self.S, frames, out, F, self.mode, threads)
        else:
            self.lib.asp_ns_oracle_run(self.states, self.S, frames, out, F, self.mode)
        return out

    def analyze(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        for i in range(self.S):
            self.lib.asp_ns_oracle_analyze(C.byref(self.states[i]), frames[i].copy(), self.mode)

    def process(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        out = np.empty_like(frames)
        for i in range(self.S):
            o = np.empty(self.block, np.float32)
            self.lib.asp_ns_oracle_process(C.byref(self.states[i]), frames[i].copy(), o, self.mode)
            out[i] = o
        return out

    def run_bands(self, low, high):
        """low [F][S][160], high [F][nh][S][160] -> (out_low, out_high): Analyze(low) +
        Process(1 + nh bands) per frame, the order of apm_ns.cpp:69-74."""
        from audiosignalprocess_amd._abi import AspNsHbState
        low = np.ascontiguousarray(low, np.float32)
        high = np.ascontiguousarray(high, np.float32)
        F, nh = low.shape[0], high.shape[1]
        if not hasattr(self, "hb"):
            self.hb = (AspNsHbState * self.S)()
        out_low, out_high = np.empty_like(low), np.empty_like(high)
        fn = self.lib.asp_ns_oracle_process_bands
        fn.argtypes = [C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int]
        fn.restype = None
        for f in range(F):
            for i in range(self.S):
                lo = low[f, i].copy()
                self.lib.asp_ns_oracle_analyze(C.byref(self.states[i]), lo.copy(), self.mode)
                hi = np.ascontiguousarray(high[f, :, i])
                ol, oh = np.empty(BLOCKL, np.float32), np.empty((nh, BLOCKL), np.float32)
                fn(C.byref(self.states[i]), C.byref(self.hb[i]), lo, hi, nh, ol, oh, self.mode)
                out_low[f, i] = ol
                out_high[f, :, i] = oh
        return out_low, out_high

    def export_state(self, stream):
        s = AspNsState()
        C.memmove(C.byref(s), C.byref(self.states[stream]), C.sizeof(AspNsState))
        return s

    def import_state(self, stream, state):
        C.memmove(C.byref(self.states[stream]), C.byref(state), C.sizeof(AspNsState))

    def rdft256(self, rows, isgn):
        rows = np.ascontiguousarray(rows, dtype=np.float32).copy()
        for r in rows.reshape(-1, 256):
            self.lib.asp_ns_oracle_rdft256(r, isgn)
        return rows

    def rdft128(self, rows, isgn):
        rows = np.ascontiguousarray(rows, dtype=np.float32).copy()
        for r in rows.reshape(-1, 128):
            self.lib.asp_ns_oracle_rdft128(r, isgn)
        return rows


class RefNs:
    """Batch of streams driven through the compiled reference (ns_core.c)."""

    def __init__(self, num_streams, policy=1, fs=16000):
        self.lib = ref_lib()
        self.S = num_streams
        self.block = 80 if fs == 8000 else BLOCKL
        self.size = self.lib.ref_ns_sizeof()
        self.buf = C.create_string_buffer(self.size * num_streams)
        self.base = C.addressof(self.buf)
        for i in range(num_streams):
            assert self.lib.WebRtcNs_InitCore(self._p(i), fs) == 0
            assert self.lib.WebRtcNs_set_policy_core(self._p(i), policy) == 0

    def _p(self, i):
        return C.c_void_p(self.base + i * self.size)

    def run(self, frames, threads=1):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        F = frames.shape[0]
        assert frames.shape == (F, self.S, self.block)
        out = np.empty_like(frames)
        self.lib.ref_ns_run(C.c_void_p(self.base), self.S, frames, out, F, threads)
        return out

    def run_bands(self, low, high):
        """low [F][S][160], high [F][nh][S][160] through WebRtcNs_AnalyzeCore + _ProcessCore with
        1 + nh bands."""
        low = np.ascontiguousarray(low, np.float32)
        high = np.ascontiguousarray(high, np.float32)
        F, nh = low.shape[0], high.shape[1]
        out_low, out_high = np.empty_like(low), np.empty_like(high)
        fn = self.lib.ref_ns_run_bands
        fn.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int]
        fn.restype = None
        for i in range(self.S):
            lo = np.ascontiguousarray(low[:, i])
            hi = np.ascontiguousarray(high[:, :, i])
            ol, oh = np.empty_like(lo), np.empty_like(hi)
            fn(self._p(i), lo, hi, nh, ol, oh, F)
            out_low[:, i] = ol
            out_high[:, :, i] = oh
        return out_low, out_high

    def export_hb(self, stream):
        from audiosignalprocess_amd._abi import AspNsHbState
        hb = AspNsHbState()
        self.lib.ref_ns_export_hb.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.ref_ns_export_hb(self._p(stream), C.byref(hb))
        return np.ctypeslib.as_array(hb.dataBufHB).reshape(2, 256).copy()

    def export_state(self, stream):
        s = AspNsState()
        self.lib.ref_ns_export(self._p(stream), C.byref(s))
        return s

    def import_state(self, stream, state):
        self.lib.ref_ns_import(self._p(stream), C.byref(state))

    def rdft256(self, rows, isgn):
        rows = np.ascontiguousarray(rows, dtype=np.float32).copy()
        for r in rows.reshape(-1, 256):
            self.lib.ref_rdft256(r, isgn)
        return rows

    def rdft128(self, rows, isgn):
        rows = np.ascontiguousarray(rows, dtype=np.float32).copy()
        for r in rows.reshape(-1, 128):
            self.lib.ref_rdft128(r, isgn)
        return rows

    def fft_tables(self):
        ip = np.zeros(128, np.int32)
        w = np.zeros(128, np.float32)
        self.lib.ref_ns_fft_tables(self._p(0), ip.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
        return ip, w


# ------------------------------------------------------------ BlockThresholding
from audiosignalprocess_amd._abi import AspBtState  # noqa: E402


_bt_ready = False
_bt_lock = __import__("threading").Lock()


def _bt_lib():
    """Prototypes are declared once, under a lock: bench.py builds one OracleBt per worker thread, and
    re-assigning argtypes while another thread is inside a call makes ctypes reject that call."""
    global _bt_ready
    lib = oracle_lib()
    with _bt_lock:
        if not _bt_ready:
            lib.bt_oracle_create.restype = C.c_void_p
            lib.bt_oracle_create.argtypes = [C.c_int]
            for n in ("bt_oracle_free", "bt_oracle_reset"):
                getattr(lib, n).argtypes = [C.c_void_p]
            lib.bt_oracle_denoise_float.argtypes = [C.c_void_p, _f32p, C.c_int]
            lib.bt_oracle_output_float.argtypes = [C.c_void_p, _f32p, C.c_int]
            lib.bt_oracle_flush_float.argtypes = [C.c_void_p, _f32p, C.c_int]
            lib.bt_oracle_macroblock.argtypes = [C.c_void_p, _f32p, _f32p, C.c_void_p]
            lib.bt_oracle_export.argtypes = [C.c_void_p, C.POINTER(AspBtState)]
            lib.bt_oracle_import.argtypes = [C.c_void_p, C.POINTER(AspBtState)]
            lib.bt_oracle_kiss_fftr.argtypes = [C.c_void_p, _f32p, _f32p]
            lib.bt_oracle_kiss_fftri.argtypes = [C.c_void_p, _f32p, _f32p]
            lib.bt_oracle_hann.restype = C.POINTER(C.c_float)
            lib.bt_oracle_hann.argtypes = [C.c_void_p]
            lib.bt_oracle_s16_to_float.restype = C.c_float
            lib.bt_oracle_s16_to_float.argtypes = [C.c_int16]
            lib.bt_oracle_float_to_s16.restype = C.c_int16
            lib.bt_oracle_float_to_s16.argtypes = [C.c_float]
            _bt_ready = True
    return lib


class OracleBt:
    """One stream through the CPU restatement of Denoise/BlockThresholding."""

    def __init__(self, win_size):
        lib = _bt_lib()
        self.lib = lib
        self.h = lib.bt_oracle_create(win_size)
        if not self.h:
            raise ValueError("unsupported win_size %d" % win_size)
        self.win = win_size
        self.half = win_size // 2
        self.macro = 8 * self.half

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.bt_oracle_free(self.h)
            self.h = None

    def macroblock(self, x, want_seg=False):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.macro,)
        y = np.empty_like(x)
        ncol = (self.win - 1) // 2 // 16
        seg = np.zeros(2 * ncol, np.int32)
        self.lib.bt_oracle_macroblock(self.h, x, y, seg.ctypes.data_as(C.c_void_p))
        return (y, seg.reshape(ncol, 2)) if want_seg else y

    def run(self, x):
        """x [num_samples] (multiple of macro) -> denoised, same length."""
        x = np.ascontiguousarray(x, np.float32)
        return np.concatenate([self.macroblock(b) for b in x.reshape(-1, self.macro)])

    def denoise_float(self, hop):
        return self.lib.bt_oracle_denoise_float(self.h, np.ascontiguousarray(hop, np.float32), len(hop))

    def output_float(self, n=None):
        out = np.zeros(self.macro if n is None else n, np.float32)
        got = self.lib.bt_oracle_output_float(self.h, out, out.size)
        return got, out

    def flush_float(self, n):
        out = np.zeros(max(n, 1), np.float32)
        got = self.lib.bt_oracle_flush_float(self.h, out, n)
        return got, out[:max(got, 0)]

    def export_state(self):
        s = AspBtState()
        self.lib.bt_oracle_export(self.h, C.byref(s))
        return s

    def import_state(self, s):
        self.lib.bt_oracle_import(self.h, C.byref(s))

    def kiss_fftr(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(self.win + 2, np.float32)
        self.lib.bt_oracle_kiss_fftr(self.h, x, out)
        return out

    def kiss_fftri(self, f):
        f = np.ascontiguousarray(f, np.float32)
        out = np.empty(self.win, np.float32)
        self.lib.bt_oracle_kiss_fftri(self.h, f, out)
        return out

    def hann(self):
        return np.ctypeslib.as_array(self.lib.bt_oracle_hann(self.h), shape=(self.win,)).copy()


# ------------------------------------------------------------------------- AEC
AEC_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libaec_ref.so")
_aec_ref = None


def have_aec_ref():
    return os.path.exists(AEC_REF_SO)


class RefAec:
    """One stream through the compiled reference echo canceller (plain-C path)."""

    def __init__(self, fs=16000):
        global _aec_ref
        if _aec_ref is None:
            lib = C.CDLL(AEC_REF_SO)
            lib.ref_aec_create.restype = C.c_void_p
            lib.ref_aec_create.argtypes = [C.c_int32]
            lib.ref_aec_free.argtypes = [C.c_void_p]
            lib.ref_aec_run.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int16]
            lib.ref_aec_frame.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int16]
            lib.ref_aec_set_nlp.argtypes = [C.c_void_p, C.c_int]
            lib.ref_aec_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            lib.aec_rdft_forward_128.argtypes = [_f32p]
            lib.aec_rdft_inverse_128.argtypes = [_f32p]
            _aec_ref = lib
        self.lib = _aec_ref
        self.h = self.lib.ref_aec_create(fs)
        assert self.h

    def run(self, far, near, delay_ms=0):
        far = np.ascontiguousarray(far, np.float32)
        near = np.ascontiguousarray(near, np.float32)
        assert far.shape == near.shape and far.shape[1] == BLOCKL
        out = np.empty_like(near)
        self.lib.ref_aec_run(self.h, far, near, out, far.shape[0], delay_ms)
        return out

    def frame(self, far, near, delay_ms=0):
        """One BufferFarend + Process of n = 80 or 160 samples; returns (out, rc)."""
        far = np.ascontiguousarray(far, np.float32)
        near = np.ascontiguousarray(near, np.float32)
        out = np.empty_like(near)
        rc = self.lib.ref_aec_frame(self.h, far, near, out, far.size, delay_ms)
        return out, rc

    def set_nlp(self, mode, metrics=0):
        self.lib.ref_aec_set_config.argtypes = [C.c_void_p, C.c_int, C.c_int]
        return self.lib.ref_aec_set_config(self.h, mode, metrics)

    def set_config(self, mode, metrics=0, skew=0, delay_logging=0):
        self.lib.ref_aec_set_config_full.argtypes = [C.c_void_p] + [C.c_int] * 4
        return self.lib.ref_aec_set_config_full(self.h, mode, metrics, skew, delay_logging)

    def enable_reported_delay(self, enable=1):
        self.lib.ref_aec_enable_reported_delay.argtypes = [C.c_void_p, C.c_int]
        self.lib.ref_aec_enable_reported_delay.restype = None
        self.lib.ref_aec_enable_reported_delay(self.h, enable)

    def delay_state(self):
        from audiosignalprocess_amd._abi import AspAecDelayState
        d = AspAecDelayState()
        self.lib.ref_aec_export_delay.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.ref_aec_export_delay.restype = None
        self.lib.ref_aec_export_delay(self.h, C.byref(d))
        return d

    def delay_metrics(self):
        med, std = C.c_int(-99), C.c_int(-99)
        self.lib.ref_aec_get_delay_metrics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        rc = self.lib.ref_aec_get_delay_metrics(self.h, C.byref(med), C.byref(std))
        return rc, med.value, std.value

    def error_code(self):
        self.lib.ref_aec_error_code.argtypes = [C.c_void_p]
        return self.lib.ref_aec_error_code(self.h)

    def skew_state(self):
        skew, res, ctr = C.c_float(), C.c_int(), C.c_int()
        self.lib.ref_aec_export_skew.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        self.lib.ref_aec_export_skew.restype = None
        self.lib.ref_aec_export_skew(self.h, C.byref(skew), C.byref(res), C.byref(ctr))
        return skew.value, res.value

    def frame_skew(self, far, near, delay_ms=0, skew=0):
        far = np.ascontiguousarray(far, np.float32)
        near = np.ascontiguousarray(near, np.float32)
        out = np.empty_like(near)
        self.lib.ref_aec_frame_skew.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int16, C.c_int32]
        rc = self.lib.ref_aec_frame_skew(self.h, far, near, out, near.size, delay_ms, skew)
        return out, rc

    def enable_delay_correction(self, enable=1):
        self.lib.ref_aec_enable_delay_correction.argtypes = [C.c_void_p, C.c_int]
        self.lib.ref_aec_enable_delay_correction.restype = None
        self.lib.ref_aec_enable_delay_correction(self.h, enable)

    def metrics_state(self):
        from audiosignalprocess_amd._abi import AspAecMetricsState
        m = AspAecMetricsState()
        self.lib.ref_aec_export_metrics.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.ref_aec_export_metrics(self.h, C.byref(m))
        return m

    def get_metrics(self):
        from audiosignalprocess_amd._abi import AecMetrics
        m = AecMetrics()
        self.lib.ref_aec_get_metrics.argtypes = [C.c_void_p, C.c_void_p]
        assert self.lib.ref_aec_get_metrics(self.h, C.byref(m)) == 0
        return m

    def frame_bands(self, far, near_low, near_high, delay_ms=0):
        far = np.ascontiguousarray(far, np.float32)
        nl, nh = np.ascontiguousarray(near_low, np.float32), np.ascontiguousarray(near_high, np.float32)
        ol, oh = np.empty_like(nl), np.empty_like(nh)
        fn = self.lib.ref_aec_frame_bands
        fn.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int16]
        rc = fn(self.h, far, nl, nh, ol, oh, far.size, delay_ms)
        return ol, oh, rc

    def dbufh(self):
        out = np.zeros(128, np.float32)
        self.lib.ref_aec_export_dbufh.argtypes = [C.c_void_p, _f32p]
        self.lib.ref_aec_export_dbufh(self.h, out)
        return out

    def export(self):
        from audiosignalprocess_amd._abi import AspAecControl, AspAecState
        st, ctl = AspAecState(), AspAecControl()
        self.lib.ref_aec_export(self.h, C.byref(st), C.byref(ctl))
        return st, ctl

    def table(self, name, n):
        """A constant table of the reference by its symbol name (rdft_w, WebRtcAec_sqrtHanning ...)."""
        return np.ctypeslib.as_array((C.c_float * n).in_dll(self.lib, name)).copy()

    def rdft128(self, rows, isgn):
        rows = np.array(rows, np.float32, copy=True)
        for r in rows.reshape(-1, 128):
            (self.lib.aec_rdft_forward_128 if isgn >= 0 else self.lib.aec_rdft_inverse_128)(r)
        return rows

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_aec_free(self.h)
            self.h = None


# ------------------------------------------------------------------------------------------
# AEC restatement (oracle/aec_oracle.c)
_aec_oracle_ready = False


def _aec_lib():
    global _aec_oracle_ready
    lib = oracle_lib()
    if not _aec_oracle_ready:
        from audiosignalprocess_amd._abi import AecConfig
        lib.asp_aec_oracle_create.restype = C.c_void_p
        lib.asp_aec_oracle_free.argtypes = [C.c_void_p]
        lib.asp_aec_oracle_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.asp_aec_oracle_set_config.argtypes = [C.c_void_p, AecConfig]
        lib.asp_aec_oracle_buffer_farend.argtypes = [C.c_void_p, _f32p, C.c_int]
        lib.asp_aec_oracle_process.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int32]
        lib.asp_aec_oracle_echo_status.argtypes = [C.c_void_p]
        lib.asp_aec_oracle_error_code.argtypes = [C.c_void_p]
        lib.asp_aec_oracle_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.asp_aec_oracle_import.argtypes = [C.c_void_p, C.c_void_p]
        lib.asp_aec_oracle_run.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
        lib.asp_aec_oracle_run_mt.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                                              C.c_int32, C.c_int]
        lib.asp_aec_oracle_rdft128.argtypes = [_f32p, C.c_int]
        lib.asp_aec_oracle_table.restype = C.POINTER(C.c_float)
        lib.asp_aec_oracle_table.argtypes = [C.c_int]
        _aec_oracle_ready = True
    return lib


class OracleAec:
    """One stream through oracle/aec_oracle.c (the CPU restatement of the reference AEC)."""

    def __init__(self, fs=16000, sc_fs=48000, nlp_mode=None):
        self.lib = _aec_lib()
        self.h = self.lib.asp_aec_oracle_create()
        assert self.h
        rc = self.lib.asp_aec_oracle_init(self.h, fs, sc_fs)
        self.init_rc = rc
        if rc == 0 and nlp_mode is not None:
            assert self.set_nlp(nlp_mode) == 0

    def set_nlp(self, mode, skew=0, metrics=0, delay_logging=0):
        from audiosignalprocess_amd._abi import AecConfig
        return self.lib.asp_aec_oracle_set_config(self.h, AecConfig(mode, skew, metrics, delay_logging))

    def enable_delay_correction(self, enable=1):
        """WebRtcAec_enable_delay_correction on the core: the extended filter (32 partitions)."""
        self.lib.asp_aec_oracle_enable_delay_correction.argtypes = [C.c_void_p, C.c_int]
        self.lib.asp_aec_oracle_enable_delay_correction.restype = None
        self.lib.asp_aec_oracle_enable_delay_correction(self.h, enable)

    def enable_reported_delay(self, enable=1):
        """WebRtcAec_enable_reported_delay on the core; 0 = the delay-agnostic mode."""
        self.lib.asp_aec_oracle_enable_reported_delay.argtypes = [C.c_void_p, C.c_int]
        self.lib.asp_aec_oracle_enable_reported_delay.restype = None
        self.lib.asp_aec_oracle_enable_reported_delay(self.h, enable)

    def delay_state(self):
        from audiosignalprocess_amd._abi import AspAecDelayState
        d = AspAecDelayState()
        self.lib.asp_aec_oracle_export_delay.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.asp_aec_oracle_export_delay.restype = None
        self.lib.asp_aec_oracle_export_delay(self.h, C.byref(d))
        return d

    def delay_metrics(self):
        """(rc, median, std) of WebRtcAec_GetDelayMetrics"""
        med, std = C.c_int(-99), C.c_int(-99)
        self.lib.asp_aec_oracle_get_delay_metrics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        rc = self.lib.asp_aec_oracle_get_delay_metrics(self.h, C.byref(med), C.byref(std))
        return rc, med.value, std.value

    def skew_state(self):
        """(skew, resample) the next BufferFarend call resamples with"""
        pos, skew, res, idx = C.c_float(), C.c_float(), C.c_int(), C.c_int()
        self.lib.asp_aec_oracle_export_skew.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        self.lib.asp_aec_oracle_export_skew.restype = None
        self.lib.asp_aec_oracle_export_skew(self.h, C.byref(pos), C.byref(skew), C.byref(res), C.byref(idx))
        return skew.value, res.value

    def frame_skew(self, far, near, delay_ms=0, skew=0):
        far = np.ascontiguousarray(far, np.float32)
        near = np.ascontiguousarray(near, np.float32)
        out = np.empty_like(near)
        rc = self.lib.asp_aec_oracle_buffer_farend(self.h, far, far.size)
        rc |= self.lib.asp_aec_oracle_process(self.h, near, out, near.size, delay_ms, skew)
        return out, rc

    def metrics_state(self):
        from audiosignalprocess_amd._abi import AspAecMetricsState
        m = AspAecMetricsState()
        self.lib.asp_aec_oracle_export_metrics.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.asp_aec_oracle_export_metrics(self.h, C.byref(m))
        return m

    def get_metrics(self):
        from audiosignalprocess_amd._abi import AecMetrics
        m = AecMetrics()
        self.lib.asp_aec_oracle_get_metrics.argtypes = [C.c_void_p, C.c_void_p]
        assert self.lib.asp_aec_oracle_get_metrics(self.h, C.byref(m)) == 0
        return m

    def run(self, far, near, delay_ms=0):
        far = np.ascontiguousarray(far, np.float32)
        near = np.ascontiguousarray(near, np.float32)
        assert far.shape == near.shape
        out = np.empty_like(near)
        self.lib.asp_aec_oracle_run(self.h, far, near, out, far.shape[0], far.shape[1], delay_ms)
        return out

    def frame(self, far, near, delay_ms=0):
        far = np.ascontiguousarray(far, np.float32)
        near = np.ascontiguousarray(near, np.float32)
        out = np.empty_like(near)
        rc = self.lib.asp_aec_oracle_buffer_farend(self.h, far, far.size)
        rc |= self.lib.asp_aec_oracle_process(self.h, near, out, near.size, delay_ms, 0)
        return out, rc

    def frame_bands(self, far, near_low, near_high, delay_ms=0):
        far = np.ascontiguousarray(far, np.float32)
        nl, nh = np.ascontiguousarray(near_low, np.float32), np.ascontiguousarray(near_high, np.float32)
        ol, oh = np.empty_like(nl), np.empty_like(nh)
        fn = self.lib.asp_aec_oracle_process_bands
        fn.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int32]
        rc = self.lib.asp_aec_oracle_buffer_farend(self.h, far, far.size)
        rc |= fn(self.h, nl, nh, ol, oh, nl.size, delay_ms, 0)
        return ol, oh, rc

    def export(self):
        from audiosignalprocess_amd._abi import AspAecControl, AspAecState
        st, ctl = AspAecState(), AspAecControl()
        self.lib.asp_aec_oracle_export(self.h, C.byref(st), C.byref(ctl))
        return st, ctl

    def error_code(self):
        return self.lib.asp_aec_oracle_error_code(self.h)

    def echo_status(self):
        return self.lib.asp_aec_oracle_echo_status(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.asp_aec_oracle_free(self.h)
            self.h = None


def aec_oracle_table(which, n):
    lib = _aec_lib()
    return np.ctypeslib.as_array(lib.asp_aec_oracle_table(which), shape=(n,)).copy()


def aec_oracle_rdft128(rows, isgn):
    lib = _aec_lib()
    rows = np.array(rows, np.float32, copy=True)
    for r in rows.reshape(-1, 128):
        lib.asp_aec_oracle_rdft128(r, isgn)
    return rows


def aec_oracle_run_mt(far, near, fs=16000, delay_ms=0, threads=1):
    """far / near [F][S][n] -> out; S independent streams on `threads` pthreads."""
    lib = _aec_lib()
    far = np.ascontiguousarray(far, np.float32)
    near = np.ascontiguousarray(near, np.float32)
    out = np.empty_like(near)
    F, S, n = far.shape
    rc = lib.asp_aec_oracle_run_mt(S, far, near, out, F, n, delay_ms, fs, threads)
    assert rc == 0
    return out


# ------------------------------------------------------------------------------------------
# two-band QMF (oracle/qmf_oracle.c; reference: oracle/_ref/libspl_ref.so)
SPL_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libspl_ref.so")
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_spl_ref = None


def have_spl_ref():
    return os.path.exists(SPL_REF_SO)


class _Qmf:
    """One channel: analysis / synthesis with caller-owned states, through `fa` / `fs`."""

    def __init__(self, fa, fs):
        self.fa, self.fs = fa, fs
        self.a1, self.a2 = np.zeros(6, np.int32), np.zeros(6, np.int32)
        self.s1, self.s2 = np.zeros(6, np.int32), np.zeros(6, np.int32)

    def analysis(self, x):
        x = np.ascontiguousarray(x, np.int16)
        low, high = np.empty(x.size // 2, np.int16), np.empty(x.size // 2, np.int16)
        self.fa(x, x.size, low, high, self.a1, self.a2)
        return low, high

    def synthesis(self, low, high):
        low, high = np.ascontiguousarray(low, np.int16), np.ascontiguousarray(high, np.int16)
        out = np.empty(2 * low.size, np.int16)
        self.fs(low, high, low.size, out, self.s1, self.s2)
        return out

    def state(self):
        return np.concatenate([self.a1, self.a2, self.s1, self.s2])


def _qmf_sig(fa, fs):
    fa.argtypes = [_i16p, C.c_int, _i16p, _i16p, _i32p, _i32p]
    fs.argtypes = [_i16p, _i16p, C.c_int, _i16p, _i32p, _i32p]
    fa.restype = None
    fs.restype = None
    return fa, fs


def OracleQmf():
    lib = oracle_lib()
    return _Qmf(*_qmf_sig(lib.asp_qmf_oracle_analysis, lib.asp_qmf_oracle_synthesis))


def RefQmf():
    global _spl_ref
    if _spl_ref is None:
        _spl_ref = C.CDLL(SPL_REF_SO)
    return _Qmf(*_qmf_sig(_spl_ref.WebRtcSpl_AnalysisQMF, _spl_ref.WebRtcSpl_SynthesisQMF))


# ------------------------------------------------------------------------------------------
# push sinc resampler (oracle/sinc_oracle.c; reference: oracle/_ref/libsinc_ref.so)
SINC_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsinc_ref.so")
_sinc_ref = None


def have_sinc_ref():
    return os.path.exists(SINC_REF_SO)


class OracleSinc:
    def __init__(self, src, dst):
        lib = oracle_lib()
        lib.asp_sinc_oracle_create.restype = C.c_void_p
        lib.asp_sinc_oracle_create.argtypes = [C.c_int, C.c_int]
        lib.asp_sinc_oracle_free.argtypes = [C.c_void_p]
        lib.asp_sinc_oracle_resample_i16.argtypes = [C.c_void_p, _i16p, _i16p]
        lib.asp_sinc_oracle_resample_i16.restype = None
        lib.asp_sinc_oracle_kernel.restype = C.POINTER(C.c_float)
        lib.asp_sinc_oracle_kernel.argtypes = [C.c_void_p]
        self.lib, self.src, self.dst = lib, src, dst
        self.h = lib.asp_sinc_oracle_create(src, dst)

    def resample(self, x):
        x = np.ascontiguousarray(x, np.int16)
        assert x.size == self.src
        out = np.empty(self.dst, np.int16)
        self.lib.asp_sinc_oracle_resample_i16(self.h, x, out)
        return out

    def kernel(self):
        return np.ctypeslib.as_array(self.lib.asp_sinc_oracle_kernel(self.h), shape=(33 * 32,)).copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.asp_sinc_oracle_free(self.h)
            self.h = None


class RefSinc:
    def __init__(self, src, dst):
        global _sinc_ref
        if _sinc_ref is None:
            lib = C.CDLL(SINC_REF_SO)
            lib.ref_sinc_create.restype = C.c_void_p
            lib.ref_sinc_create.argtypes = [C.c_int, C.c_int]
            lib.ref_sinc_free.argtypes = [C.c_void_p]
            lib.ref_sinc_resample_i16.argtypes = [C.c_void_p, _i16p, C.c_int, _i16p, C.c_int]
            _sinc_ref = lib
        self.lib, self.src, self.dst = _sinc_ref, src, dst
        self.h = self.lib.ref_sinc_create(src, dst)

    def resample(self, x):
        x = np.ascontiguousarray(x, np.int16)
        out = np.empty(self.dst, np.int16)
        assert self.lib.ref_sinc_resample_i16(self.h, x, self.src, out, self.dst) == self.dst
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_sinc_free(self.h)
            self.h = None


# ------------------------------------------------------------------------------------------
# SplittingFilter (oracle/split_oracle.c; reference: oracle/_ref/libsplit_ref.so)
SPLIT_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsplit_ref.so")
_split_ref = None


def have_split_ref():
    return os.path.exists(SPLIT_REF_SO)


class OracleSplit:
    def __init__(self, num_bands):
        lib = oracle_lib()
        lib.asp_split_oracle_create.restype = C.c_void_p
        lib.asp_split_oracle_create.argtypes = [C.c_int]
        lib.asp_split_oracle_free.argtypes = [C.c_void_p]
        lib.asp_split_oracle_analysis.argtypes = [C.c_void_p, _i16p, _i16p]
        lib.asp_split_oracle_synthesis.argtypes = [C.c_void_p, _i16p, _i16p]
        lib.asp_split_oracle_analysis.restype = None
        lib.asp_split_oracle_synthesis.restype = None
        self.lib, self.nb = lib, num_bands
        self.h = lib.asp_split_oracle_create(num_bands)

    def analysis(self, x):
        x = np.ascontiguousarray(x, np.int16)
        bands = np.empty((self.nb, 160), np.int16)
        self.lib.asp_split_oracle_analysis(self.h, x, bands)
        return bands

    def synthesis(self, bands):
        bands = np.ascontiguousarray(bands, np.int16)
        out = np.empty(160 * self.nb, np.int16)
        self.lib.asp_split_oracle_synthesis(self.h, bands, out)
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.asp_split_oracle_free(self.h)
            self.h = None


class RefSplit:
    def __init__(self, num_bands):
        global _split_ref
        if _split_ref is None:
            lib = C.CDLL(SPLIT_REF_SO)
            lib.ref_split_create.restype = C.c_void_p
            lib.ref_split_create.argtypes = [C.c_int, C.c_int]
            lib.ref_split_free.argtypes = [C.c_void_p]
            lib.ref_split_analysis.argtypes = [C.c_void_p, _i16p, C.c_int, C.c_int, _i16p]
            lib.ref_split_synthesis.argtypes = [C.c_void_p, _i16p, C.c_int, C.c_int, _i16p]
            _split_ref = lib
        self.lib, self.nb = _split_ref, num_bands
        self.h = self.lib.ref_split_create(160 * num_bands, num_bands)

    def analysis(self, x):
        x = np.ascontiguousarray(x, np.int16)
        bands = np.empty((self.nb, 160), np.int16)
        self.lib.ref_split_analysis(self.h, x, 160 * self.nb, self.nb, bands)
        return bands

    def synthesis(self, bands):
        bands = np.ascontiguousarray(bands, np.int16)
        out = np.empty(160 * self.nb, np.int16)
        self.lib.ref_split_synthesis(self.h, bands, 160 * self.nb, self.nb, out)
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_split_free(self.h)
            self.h = None


# ------------------------------------------------------------------------------------------
# libapm APM_NS class end to end (reference: oracle/_ref/libapm_ref.so)
APM_REF_SO = os.path.join(ORACLE_DIR, "_ref", "libapm_ref.so")


def have_apm_ref():
    return os.path.exists(APM_REF_SO)


class RefApm:
    def __init__(self, freq, mode, channels):
        lib = C.CDLL(APM_REF_SO)
        lib.ref_apm_create.restype = C.c_void_p
        lib.ref_apm_create.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int]
        lib.ref_apm_free.argtypes = [C.c_void_p]
        lib.ref_apm_process_s16.argtypes = [C.c_void_p, _i16p, C.c_int, C.c_int]
        lib.ref_apm_process_f32.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        self.lib, self.ch, self.spc = lib, channels, freq // 100
        self.h = lib.ref_apm_create(freq, mode, self.spc, channels)
        assert self.h

    def process_f32(self, frame):
        """frame [spc][channels] float32 interleaved in [-1, 1]; returns the processed copy."""
        x = np.array(frame, np.float32, copy=True)
        self.lib.ref_apm_process_f32(self.h, x.reshape(-1), self.spc, self.ch)
        return x

    def process_s16(self, frame):
        x = np.array(frame, np.int16, copy=True)
        self.lib.ref_apm_process_s16(self.h, x.reshape(-1), self.spc, self.ch)
        return x

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_apm_free(self.h)
            self.h = None
