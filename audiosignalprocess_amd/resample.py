"""Python mirror of the batched push sinc resampler C-ABI (include/asp_resample.h) over ctypes.
Plumbing only -- every call goes into libasp_amd.so; no CPU fallback."""
import ctypes as C

import numpy as np

from ._abi import MEM_HOST
from .ns import AspError, load_library

_sig_done = False


def _lib():
    global _sig_done
    lib = load_library()
    if not _sig_done:
        vp, ip = C.c_void_p, C.c_int
        sig = {
            "AspSincBatch_Create": [C.POINTER(vp), ip, ip, ip, ip],
            "AspSincBatch_Free": [vp],
            "AspSincBatch_num_channels": [vp],
            "AspSincBatch_Resample": [vp, vp, vp, ip],
            "AspSincBatch_Synchronize": [vp],
            "AspSincBatch_kernel_table": [vp, vp, ip],
        }
        for name, args in sig.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _sig_done = True
    return lib


class SincBatch:
    """N independent channels of PushSincResampler(src_frames, dst_frames) on one GPU."""

    def __init__(self, num_channels, src_frames, dst_frames, device=0):
        self.lib = _lib()
        self.C, self.src, self.dst = int(num_channels), int(src_frames), int(dst_frames)
        h = C.c_void_p()
        rc = self.lib.AspSincBatch_Create(C.byref(h), self.C, self.src, self.dst, device)
        if rc != 0:
            raise AspError("AspSincBatch_Create failed (%d)" % rc)
        self.h = h

    def resample(self, x):
        x = np.ascontiguousarray(x, np.int16)
        assert x.shape == (self.C, self.src)
        out = np.empty((self.C, self.dst), np.int16)
        rc = self.lib.AspSincBatch_Resample(self.h, x.ctypes.data, out.ctypes.data, MEM_HOST)
        if rc != 0:
            raise AspError("AspSincBatch_Resample failed (%d)" % rc)
        return out

    def kernel_table(self):
        buf = np.zeros(33 * 32, np.float32)
        assert self.lib.AspSincBatch_kernel_table(self.h, buf.ctypes.data, buf.size) == buf.size
        return buf

    def close(self):
        if getattr(self, "h", None):
            self.lib.AspSincBatch_Free(self.h)
            self.h = None

    def __del__(self):
        self.close()
