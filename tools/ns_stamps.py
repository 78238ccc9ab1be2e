import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosignalprocess_amd.ns import NsBatch
from audiosignalprocess_amd.synth import ns_frames
names = ["in+energy", "fftF", "g2loads+magn+log", "sums1", "trackers", "startup", "snr", "flat+diff", "hist", "speechprob", "noiseupd", "gain", "ifft", "gainfac", "ola", "scalars"]
for S in (8, 4096):
    g = NsBatch(S, policy=1)
    x = torch.from_numpy(ns_frames(S, 260, frame0=0)).cuda()
    y = torch.empty_like(x)
    g.analyze_process_device(x.data_ptr(), y.data_ptr(), 259)
    g.synchronize()
    st = (C.c_ulonglong * 16)()
    g.lib.AspNsBatch_DebugStamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert g.lib.AspNsBatch_DebugStamps(g.h, C.c_void_p(x[259].data_ptr()), C.c_void_p(y[259].data_ptr()), st) == 0
    t = np.array(list(st), dtype=np.int64)
    d = np.diff(t)
    print("S=%d total %d:" % (S, t[-1] - t[0]), {n: int(v) for n, v in zip(names, d)})
