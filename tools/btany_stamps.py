"""Phase stamps (s_memtime ticks of workgroup 0) of bt_macroblock_any_kernel for a few windows."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiosignalprocess_amd.bt import BtBatch
from audiosignalprocess_amd.ns import DeviceBuffer
from audiosignalprocess_amd.synth import bt_samples
names = ["load+perm", "fftF", "split", "sq", "SURE", "dc+argmin+stein", "wiener", "merge", "fftI", "ola"]
for n in (320, 960):
    for S in (1, 4096):
        g = BtBatch(S, n)
        x = bt_samples(S, g.macro)
        dx, dy = DeviceBuffer(x.nbytes), DeviceBuffer(x.nbytes)
        dx.upload(x)
        st = (C.c_ulonglong * 48)()
        g.lib.AspBtBatch_DebugStamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for rep in range(2):
            assert g.lib.AspBtBatch_DebugStamps(g.h, C.c_void_p(dx.ptr), C.c_void_p(dy.ptr), st) == 0
        t = np.array(list(st), dtype=np.int64)[:11]
        print("win %d S=%d total %d ticks:" % (n, S, t[-1] - t[0]), {k: int(v) for k, v in zip(names, np.diff(t))})
        g.close()
