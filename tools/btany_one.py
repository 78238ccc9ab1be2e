"""One window of the mixed-radix BlockThresholding path, a few K-step regions (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiosignalprocess_amd import bt
from audiosignalprocess_amd.ns import DeviceBuffer
from audiosignalprocess_amd.synth import bt_samples
n = int(sys.argv[1]) if len(sys.argv) > 1 else 960
S = 4096
g = bt.BtBatch(S, n)
x = bt_samples(S, 2 * g.macro).reshape(S, 2, g.macro).transpose(1, 0, 2).copy()
dx, dy = DeviceBuffer(x.nbytes), DeviceBuffer(x.nbytes)
dx.upload(x)
os.environ["ASP_BT_CHAINS"] = "1"
g.timed_steps(dx.ptr, dy.ptr, 2, 4)
print("win %d: %.1f us" % (n, 1000 * g.timed_steps(dx.ptr, dy.ptr, 2, 10) / 10))
