"""Phase stamps of one wave of the hand-off build in the middle of a 20-step launch (steady state), beside
the plain build's (one launch per step, a wave of workgroup 0).  Shader-clock ticks between the marks."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosignalprocess_amd.ns import NsBatch
from audiosignalprocess_amd.synth import ns_frames
names = ["in+energy", "fftF", "g2loads+magn+log", "sums1", "trackers", "startup", "snr", "flat+diff", "hist", "speechprob",
         "noiseupd", "gain", "ifft", "gainfac", "ola", "scalars", "drain"]
for S in [int(a) for a in sys.argv[1:]] or [4096, 8192]:
    g = NsBatch(S, policy=1)
    ring = 20
    x = torch.from_numpy(ns_frames(S, ring, frame0=50)).cuda()
    y = torch.empty_like(x)
    for _ in range(14):
        g.analyze_process_device(x.data_ptr(), y.data_ptr(), ring)
    g.synchronize()
    st = (C.c_ulonglong * 17)()
    g.lib.AspNsBatch_DebugFlowStamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for rep in range(3):
        rc = g.lib.AspNsBatch_DebugFlowStamps(g.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), ring, 20, st)
        assert rc == 0, rc
        t = np.array(list(st), dtype=np.int64)
        d = np.diff(t)
        print("flow  S=%d total %d:" % (S, t[-1] - t[0]), {n: int(v) for n, v in zip(names, d)})
    st16 = (C.c_ulonglong * 16)()
    g.lib.AspNsBatch_DebugStamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for rep in range(2):
        assert g.lib.AspNsBatch_DebugStamps(g.h, C.c_void_p(x[0].data_ptr()), C.c_void_p(y[0].data_ptr()), st16) == 0
        t = np.array(list(st16), dtype=np.int64)
        print("plain S=%d total %d:" % (S, t[-1] - t[0]), {n: int(v) for n, v in zip(names, np.diff(t))})
    g.close()
