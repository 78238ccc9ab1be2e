"""Where the mixed-radix BT step goes: 7-hop flushes (STFT + inverse STFT + overlap-add only) against whole
macroblocks (the same + SURE, attenuation, Wiener), device buffers, wall clock over queued calls."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiosignalprocess_amd import bt
from audiosignalprocess_amd.ns import DeviceBuffer
from audiosignalprocess_amd.synth import bt_samples
S = 4096
for n in (320, 960, 256, 1024):
    g = bt.BtBatch(S, n)
    x = bt_samples(S, g.macro)
    dx, dy = DeviceBuffer(x.nbytes), DeviceBuffer(x.nbytes)
    dx.upload(x)
    lib = g.lib
    def run(fn, reps=30):
        for _ in range(5): fn()
        g.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        g.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    den = run(lambda: lib.AspBtBatch_Denoise(g.h, C.c_void_p(dx.ptr), C.c_void_p(dy.ptr), 1))
    flu = run(lambda: lib.AspBtBatch_Flush(g.h, C.c_void_p(dx.ptr), 7, C.c_void_p(dy.ptr), 1))
    print("win %4d: macroblock %.1f us, 7-hop flush (no thresholding) %.1f us" % (n, den, flu))
    g.close()
