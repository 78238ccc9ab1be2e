"""Python mirror of the batched BlockThresholding C-ABI (include/asp_bt.h) over ctypes.
Plumbing only -- every call goes into libasp_amd.so; no CPU fallback."""
import ctypes as C

import numpy as np

from ._abi import MEM_DEVICE, MEM_HOST, AspBtState
from .ns import AspError, load_library

_sig_done = False


def _lib():
    global _sig_done
    lib = load_library()
    if not _sig_done:
        vp, ip = C.c_void_p, C.c_int
        sig = {
            "AspBtBatch_Create": [C.POINTER(vp), ip, ip, ip],
            "AspBtBatch_Free": [vp],
            "AspBtBatch_Reset": [vp],
            "AspBtBatch_ResetStream": [vp, C.c_int],
            "AspBtBatch_DenoiseBlocks": [vp, vp, vp, C.c_int, C.c_int],
            "AspBtBatch_SetFlow": [vp, C.c_int],
            "AspBtBatch_num_streams": [vp],
            "AspBtBatch_macro_size": [vp],
            "AspBtBatch_Denoise": [vp, vp, vp, ip],
            "AspBtBatch_Flush": [vp, vp, ip, vp, ip],
            "AspBtBatch_ExportState": [vp, ip, C.POINTER(AspBtState)],
            "AspBtBatch_ImportState": [vp, ip, C.POINTER(AspBtState)],
            "AspBtBatch_Synchronize": [vp],
            "AspBtBatch_TimedSteps": [vp, vp, vp, ip, ip, C.POINTER(C.c_float)],
            "AspBt_kiss_fftr_batch": [vp, vp, ip, ip, ip],
            "AspBt_kiss_fftri_batch": [vp, vp, ip, ip, ip],
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


class BtBatch:
    """N independent stream-channels of the block-thresholding denoiser on one GPU."""

    def __init__(self, num_streams, win_size=1024, device=0):
        self.lib = _lib()
        self.S, self.win = int(num_streams), int(win_size)
        self.half, self.macro = self.win // 2, 4 * self.win
        h = C.c_void_p()
        _check(self.lib.AspBtBatch_Create(C.byref(h), self.S, self.win, device), "AspBtBatch_Create")
        self.h = h

    def denoise(self, x):
        """x [S][macro] float32 (one macroblock per stream) -> [S][macro]."""
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.S, self.macro)
        y = np.empty_like(x)
        _check(self.lib.AspBtBatch_Denoise(self.h, x.ctypes.data, y.ctypes.data, MEM_HOST),
               "AspBtBatch_Denoise")
        return y

    def run(self, x):
        """x [S][k*macro] -> denoised [S][k*macro] (k sequential macroblocks)."""
        x = np.ascontiguousarray(x, np.float32)
        k = x.shape[1] // self.macro
        return np.concatenate([self.denoise(x[:, i * self.macro:(i + 1) * self.macro]) for i in range(k)], axis=1)

    def flush(self, x, hops):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.S, hops * self.half)
        y = np.empty_like(x)
        _check(self.lib.AspBtBatch_Flush(self.h, x.ctypes.data, hops, y.ctypes.data, MEM_HOST),
               "AspBtBatch_Flush")
        return y

    def denoise_device(self, in_ptr, out_ptr):
        _check(self.lib.AspBtBatch_Denoise(self.h, C.c_void_p(in_ptr), C.c_void_p(out_ptr), MEM_DEVICE),
               "AspBtBatch_Denoise")

    def timed_steps(self, in_ptr, out_ptr, blocks_in_ring, steps):
        ms = C.c_float()
        _check(self.lib.AspBtBatch_TimedSteps(self.h, C.c_void_p(in_ptr), C.c_void_p(out_ptr),
                                              blocks_in_ring, steps, C.byref(ms)), "AspBtBatch_TimedSteps")
        return ms.value

    def reset(self):
        _check(self.lib.AspBtBatch_Reset(self.h), "AspBtBatch_Reset")

    def set_flow(self, mode):
        """Hand-off build of denoise_blocks() / timed_steps(): -1 default, 0 off, 1 on (include/asp_bt.h)."""
        _check(self.lib.AspBtBatch_SetFlow(self.h, mode), "AspBtBatch_SetFlow")

    def denoise_blocks(self, x):
        """x [K][S][macro] -> denoised, K consecutive macroblocks of every stream-channel in one call."""
        x = np.ascontiguousarray(x, np.float32)
        assert x.ndim == 3 and x.shape[1:] == (self.S, self.macro)
        y = np.empty_like(x)
        _check(self.lib.AspBtBatch_DenoiseBlocks(self.h, x.ctypes.data, y.ctypes.data, x.shape[0], MEM_HOST),
               "AspBtBatch_DenoiseBlocks")
        return y

    def reset_stream(self, stream):
        """blockThreshold_reset of one stream-channel of the running batch."""
        _check(self.lib.AspBtBatch_ResetStream(self.h, stream), "AspBtBatch_ResetStream")

    def synchronize(self):
        _check(self.lib.AspBtBatch_Synchronize(self.h), "AspBtBatch_Synchronize")

    def export_state(self, stream):
        s = AspBtState()
        _check(self.lib.AspBtBatch_ExportState(self.h, stream, C.byref(s)), "AspBtBatch_ExportState")
        return s

    def import_state(self, stream, s):
        _check(self.lib.AspBtBatch_ImportState(self.h, stream, C.byref(s)), "AspBtBatch_ImportState")

    def close(self):
        if getattr(self, "h", None):
            self.lib.AspBtBatch_Free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kiss_fftr(x, n):
    """kiss_fftr on every row of x [count][n] -> [count][n+2] interleaved {r, i}."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, n)
    out = np.empty((x.shape[0], n + 2), np.float32)
    _check(_lib().AspBt_kiss_fftr_batch(x.ctypes.data, out.ctypes.data, n, x.shape[0], 0), "kiss_fftr")
    return out


def kiss_fftri(f, n):
    f = np.ascontiguousarray(f, np.float32).reshape(-1, n + 2)
    out = np.empty((f.shape[0], n), np.float32)
    _check(_lib().AspBt_kiss_fftri_batch(f.ctypes.data, out.ctypes.data, n, f.shape[0], 0), "kiss_fftri")
    return out
