import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosignalprocess_amd.bt import BtBatch
from audiosignalprocess_amd.synth import bt_samples
for S in (1, 512, 4096):
    g = BtBatch(S, 1024)
    x = torch.from_numpy(bt_samples(S, g.macro)).cuda()
    y = torch.empty_like(x)
    st = (C.c_ulonglong * 48)()
    g.lib.AspBtBatch_DebugStamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for rep in range(2):
        assert g.lib.AspBtBatch_DebugStamps(g.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), st) == 0
    t = np.array(list(st), dtype=np.int64)
    names = ["load+tables", "fftF", "splitF", "barrier1", "SURE", "barrier2", "stein+wiener", "barrier3", "merge+fftI", "ola"]
    wv = t[16:48].reshape(8, 4); sub = t[11:16]; t = t[:11]
    d = np.diff(t)
    print("S=%d total %d ticks:" % (S, t[-1] - t[0]), {n: int(v) for n, v in zip(names, d)}, "sub-stamps since phase 6 start:", [int(v - t[6]) for v in sub if v])
    if wv.any():
        print("   per wave, ticks since kernel start [end of phase A, end of SURE, end of B2, end]:", [[int(v - t[0]) for v in r] for r in wv])
