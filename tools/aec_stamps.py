import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosignalprocess_amd.aec import AecBatch
from audiosignalprocess_amd.synth import aec_frames
names = ["loads", "nearfft", "power", "filterfar", "ifft+e", "Efft", "scale", "adapt", "partdelay", "winffts",
         "psd+coh", "nlp scalars", "overdrive+noise", "ifft+ola", "carry"]
for S in (4, 4096):
    g = AecBatch(S)
    far, near = aec_frames(4, 90)
    idx = np.arange(S) % 4
    tf = torch.from_numpy(np.ascontiguousarray(far[:, idx])).cuda()
    tn = torch.from_numpy(np.ascontiguousarray(near[:, idx])).cuda()
    to = torch.empty_like(tn)
    per = S * 160 * 4
    g.run_device(tf.data_ptr(), tn.data_ptr(), to.data_ptr(), 160, 80)
    g.synchronize()
    st = (C.c_ulonglong * 16)()
    g.lib.AspAecBatch_DebugStamps.argtypes = [C.c_void_p] * 5
    for f in range(80, 84):
        rc = g.lib.AspAecBatch_DebugStamps(g.h, C.c_void_p(tf.data_ptr() + f * per), C.c_void_p(tn.data_ptr() + f * per),
                                           C.c_void_p(to.data_ptr() + f * per), st)
        assert rc == 0
        t = np.array(list(st), dtype=np.int64)
        entry = t[4]              # mark 4: the call's entry, ahead of the fused far-end work (process_call)
        t[4] = t[3]
        d = np.diff(t)
        print("S=%d f=%d first block %d ticks, far-end work + sub-frame set-up ahead of it %d ticks:" % (S, f, t[-1] - t[0], t[0] - entry),
              {n: int(v) for n, v in zip(names, d)})
