"""Step time of the mixed-radix BlockThresholding path (bt_macroblock_any_kernel) for a few windows, 4096
stream-channels, one macroblock per stream and step (hipEvent time of AspBtBatch_TimedSteps)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiosignalprocess_amd import bt
from audiosignalprocess_amd.ns import DeviceBuffer
from audiosignalprocess_amd.synth import bt_samples

S = 4096
for n in (256, 320, 480, 960, 1024, 1920):
    g = bt.BtBatch(S, n)
    x = bt_samples(S, 2 * g.macro).reshape(S, 2, g.macro).transpose(1, 0, 2).copy()
    dx, dy = DeviceBuffer(x.nbytes), DeviceBuffer(x.nbytes)
    dx.upload(x)
    g.timed_steps(dx.ptr, dy.ptr, 2, 4)
    ms = g.timed_steps(dx.ptr, dy.ptr, 2, 20)
    us = 1000 * ms / 20
    print("win %4d: %7.1f us per step of %d macroblocks = %.2f G samples/s, %.3f of 8 TB/s on 10 B / sample"
          % (n, us, S, S * g.macro / us / 1e3, S * g.macro * 10 / (us * 1e-6) / 8e12))
    g.close()
