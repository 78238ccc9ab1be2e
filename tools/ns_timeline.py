"""Launch-level timeline of the fused NS step (pair-layout kernel): every workgroup's first wave stamps the
100 MHz real-time counter at its start, after its first loads, before its last stores and at its end
(AspNsBatch_DebugTimeline); printed per launch chain for the last of `steps` back-to-back steps.
usage: python3 tools/ns_timeline.py [streams] [split] [steps]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiosignalprocess_amd.ns import NsBatch  # noqa: E402
from audiosignalprocess_amd.synth import ns_frames  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
split = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ring = 50
g = NsBatch(S, policy=1)
g.set_split(split)
base = ns_frames(16, 260 + ring, frame0=0)
x = torch.from_numpy(np.ascontiguousarray(base[:, np.arange(S) % 16])).cuda()
y = torch.empty_like(x)
g.analyze_process_device(x.data_ptr(), y.data_ptr(), 260)  # past both start-up windows
g.synchronize()
nwg = S // 4
buf = (C.c_ulonglong * (nwg * 4))()
fn = g.lib.AspNsBatch_DebugTimeline
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
for rep in range(2):
    rc = fn(g.h, C.c_void_p(x[260].data_ptr()), C.c_void_p(y.data_ptr()), ring, steps, buf, nwg)
    assert rc == 0, g.lib.AspNs_last_error()
    raw = np.array(list(buf), dtype=np.uint64).reshape(nwg, 4)
    cu = (raw[:, 0] >> np.uint64(48)).astype(np.int64)
    raw[:, 0] &= np.uint64((1 << 48) - 1)
    t = raw.astype(np.float64) * 0.01  # us
    t0 = t[:, 0].min()
    t -= t0
    print("S=%d split=%d steps=%d (rep %d): last step, us relative to the earliest workgroup start of that step" % (S, split, steps, rep))
    bounds = [int((S * p // split) // 8 * 8) // 4 for p in range(split)] + [nwg]
    for c in range(split):
        w = t[bounds[c]:bounds[c + 1]]
        q = lambda a: "min %6.2f p10 %6.2f med %6.2f p90 %6.2f max %6.2f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
        print(" chain %d (%d workgroups)" % (c, len(w)))
        print("   start        ", q(w[:, 0]))
        print("   loads in     ", q(w[:, 1]))
        print("   before stores", q(w[:, 2]))
        print("   end          ", q(w[:, 3]))
        print("   wave: load wait", q(w[:, 1] - w[:, 0]), "| compute", q(w[:, 2] - w[:, 1]), "| total", q(w[:, 3] - w[:, 0]))
        print("   launch span (first start -> last end): %.2f us" % (w[:, 3].max() - w[:, 0].min()))
        cc = cu[bounds[c]:bounds[c + 1]]
        ids, cnt = np.unique(cc, return_counts=True)
        comp = w[:, 2] - w[:, 1]
        print("   workgroups per CU: %d CUs used; histogram %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
        for n in np.unique(cnt):
            sel = np.isin(cc, ids[cnt == n])
            print("     CUs with %d workgroups of this chain: compute med %.2f max %.2f us" % (n, np.median(comp[sel]), comp[sel].max()))
    ids, cnt = np.unique(cu, return_counts=True)
    print(" all chains: %d CUs used; workgroups per CU histogram %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
g.close()
