"""PCIe-inclusive NS rate: host frames in, host frames out (ASP_MEM_HOST), 4096 streams."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiosignalprocess_amd.ns import NsBatch
from audiosignalprocess_amd.synth import ns_frames
S, F = 4096, 50
x = ns_frames(S, F, frame0=50)
g = NsBatch(S, policy=1)
g.set_split(2)
for _ in range(5):
    g.analyze_process(x)          # warm-up: 250 frames
t0 = time.perf_counter()
reps = 6
for _ in range(reps):
    g.analyze_process(x)
dt = time.perf_counter() - t0
print("PCIe-inclusive: %.1f M frames/s (%d streams, %d-frame calls, pageable host buffers, %.2f GB/s each way)"
      % (S * F * reps / dt / 1e6, S, F, S * F * reps * 640 / dt / 1e9))
