"""Python mirror of the batched two-band QMF C-ABI (include/asp_split.h) over ctypes.
Plumbing only -- every call goes into libasp_amd.so; no CPU fallback."""
import ctypes as C

import numpy as np

from ._abi import MEM_DEVICE, MEM_HOST
from .ns import AspError, load_library

_sig_done = False


class AspQmfState(C.Structure):
    """include/asp_split.h: AspQmfState (TwoBandsStates of one channel)."""

    _fields_ = [("analysis_state1", C.c_int32 * 6), ("analysis_state2", C.c_int32 * 6),
                ("synthesis_state1", C.c_int32 * 6), ("synthesis_state2", C.c_int32 * 6)]


def _lib():
    global _sig_done
    lib = load_library()
    if not _sig_done:
        vp, ip = C.c_void_p, C.c_int
        sig = {
            "AspQmfBatch_Create": [C.POINTER(vp), ip, ip],
            "AspQmfBatch_Free": [vp],
            "AspQmfBatch_Reset": [vp],
            "AspQmfBatch_num_channels": [vp],
            "AspQmfBatch_Analysis": [vp, vp, ip, vp, vp, ip],
            "AspQmfBatch_Synthesis": [vp, vp, vp, ip, vp, ip],
            "AspQmfBatch_ExportState": [vp, ip, C.POINTER(AspQmfState)],
            "AspQmfBatch_ImportState": [vp, ip, C.POINTER(AspQmfState)],
            "AspQmfBatch_Synchronize": [vp],
        }
        for name, args in sig.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _sig_done = True
    return lib


def _check(rc, what):
    if rc != 0:
        raise AspError("%s failed (%d)" % (what, rc))


class QmfBatch:
    """N independent channels of the two-band split / merge on one GPU."""

    def __init__(self, num_channels, device=0):
        self.lib = _lib()
        self.C = int(num_channels)
        h = C.c_void_p()
        _check(self.lib.AspQmfBatch_Create(C.byref(h), self.C, device), "AspQmfBatch_Create")
        self.h = h

    def analysis(self, x):
        """x [C][2L] int16 -> (low [C][L], high [C][L])."""
        x = np.ascontiguousarray(x, np.int16)
        L = x.shape[1] // 2
        assert x.shape == (self.C, 2 * L)
        low, high = np.empty((self.C, L), np.int16), np.empty((self.C, L), np.int16)
        _check(self.lib.AspQmfBatch_Analysis(self.h, x.ctypes.data, L, low.ctypes.data, high.ctypes.data, MEM_HOST),
               "AspQmfBatch_Analysis")
        return low, high

    def synthesis(self, low, high):
        low, high = np.ascontiguousarray(low, np.int16), np.ascontiguousarray(high, np.int16)
        L = low.shape[1]
        assert low.shape == high.shape == (self.C, L)
        out = np.empty((self.C, 2 * L), np.int16)
        _check(self.lib.AspQmfBatch_Synthesis(self.h, low.ctypes.data, high.ctypes.data, L, out.ctypes.data, MEM_HOST),
               "AspQmfBatch_Synthesis")
        return out

    def state(self, channel):
        st = AspQmfState()
        _check(self.lib.AspQmfBatch_ExportState(self.h, channel, C.byref(st)), "AspQmfBatch_ExportState")
        return np.concatenate([np.ctypeslib.as_array(getattr(st, n)) for n, _ in st._fields_]).copy()

    def reset(self):
        _check(self.lib.AspQmfBatch_Reset(self.h), "AspQmfBatch_Reset")

    def close(self):
        if getattr(self, "h", None):
            self.lib.AspQmfBatch_Free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class SplitBatch:
    """N independent channels of the reference's SplittingFilter (2 bands at 32 kHz, 3 at 48 kHz)."""

    def __init__(self, num_channels, num_bands, device=0):
        self.lib = _lib()
        vp, ip = C.c_void_p, C.c_int
        self.lib.AspSplitBatch_Create.argtypes = [C.POINTER(vp), ip, ip, ip]
        self.lib.AspSplitBatch_Free.argtypes = [vp]
        self.lib.AspSplitBatch_Analysis.argtypes = [vp, vp, vp, ip]
        self.lib.AspSplitBatch_Synthesis.argtypes = [vp, vp, vp, ip]
        self.C, self.nb = int(num_channels), int(num_bands)
        h = C.c_void_p()
        _check(self.lib.AspSplitBatch_Create(C.byref(h), self.C, self.nb, device), "AspSplitBatch_Create")
        self.h = h

    def analysis(self, x):
        """x [C][160 nb] int16 -> bands [nb][C][160]."""
        x = np.ascontiguousarray(x, np.int16)
        assert x.shape == (self.C, 160 * self.nb)
        bands = np.empty((self.nb, self.C, 160), np.int16)
        _check(self.lib.AspSplitBatch_Analysis(self.h, x.ctypes.data, bands.ctypes.data, MEM_HOST),
               "AspSplitBatch_Analysis")
        return bands

    def synthesis(self, bands):
        bands = np.ascontiguousarray(bands, np.int16)
        assert bands.shape == (self.nb, self.C, 160)
        out = np.empty((self.C, 160 * self.nb), np.int16)
        _check(self.lib.AspSplitBatch_Synthesis(self.h, bands.ctypes.data, out.ctypes.data, MEM_HOST),
               "AspSplitBatch_Synthesis")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.AspSplitBatch_Free(self.h)
            self.h = None

    def __del__(self):
        self.close()
