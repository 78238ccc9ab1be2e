import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from audiosignalprocess_amd.aec import AecBatch
from audiosignalprocess_amd.bt import BtBatch
from audiosignalprocess_amd.synth import aec_frames
S=4; ring=40
far1, near1 = aec_frames(64, ring)
idx=np.arange(S)%64
d_far=torch.from_numpy(np.ascontiguousarray(far1[:, idx])).cuda(); d_near=torch.from_numpy(np.ascontiguousarray(near1[:, idx])).cuda(); d_out=torch.empty_like(d_near)
g=AecBatch(S,16000)
g.timed_steps(d_far.data_ptr(), d_near.data_ptr(), d_out.data_ptr(), 160, ring, 200)
for steps in (64, 256, 1024):
    t0=time.perf_counter(); ms=g.timed_steps(d_far.data_ptr(), d_near.data_ptr(), d_out.data_ptr(), 160, ring, steps); w=time.perf_counter()-t0
    print("AEC S=4 steps %d: event %.3f ms wall %.3f ms -> %.2f us per frame of host + launch time" % (steps, ms, 1e3*w, 1e3*ms/steps))
