#!/bin/bash
# Timing probe: what the second (bin 64) trip of the AEC's per-bin phases costs.  Builds aec_kernels.hip with
# AEC_TRIPS=1 (bin 64 is then not computed: the results are WRONG, only the step time means something) into a
# library of its OWN under tools/probe/bin (tools/build_variant.sh) and loads it through ASP_AMD_LIB: the in-tree
# library every test and bench run loads is never replaced, whatever happens to this script.
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/build_variant.sh trips1 aec_kernels.hip -DAEC_TRIPS=1 > gpurun_out/build_trips1.log 2>&1 || { tail -5 gpurun_out/build_trips1.log; exit 1; }
for L in "" tools/probe/bin/libasp_trips1.so ""; do
  ASP_AMD_LIB="${L:+$PWD/$L}" python3 - <<PY
import numpy as np, torch
from audiosignalprocess_amd.aec import AecBatch
from audiosignalprocess_amd.synth import aec_frames
S, ring = 4096, 40
far1, near1 = aec_frames(64, ring)
idx = np.arange(S) % 64
d_far = torch.from_numpy(np.ascontiguousarray(far1[:, idx])).cuda()
d_near = torch.from_numpy(np.ascontiguousarray(near1[:, idx])).cuda()
d_out = torch.empty_like(d_near)
g = AecBatch(S, 16000)
g.timed_steps(d_far.data_ptr(), d_near.data_ptr(), d_out.data_ptr(), 160, ring, 125)
for _ in range(2):
    ms = g.timed_steps(d_far.data_ptr(), d_near.data_ptr(), d_out.data_ptr(), 160, ring, 250)
    print("${L:-in-tree (AEC_TRIPS=2)}: step_us %.1f" % (1000 * ms / 250))
PY
done
