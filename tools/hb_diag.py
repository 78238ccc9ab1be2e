import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from audiosignalprocess_amd import ns
from tests.oracle_lib import OracleNs, REDUCE_TREE32, REDUCE_TREE
from tests.test_ns_oracle import _band_frames
S, F = 6, 60
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for fs, nh, spw in [(32000, 1, 2), (32000, 1, 1)]:
    low, high = _band_frames(S, 320, nh); low, high = np.ascontiguousarray(low[:F]), np.ascontiguousarray(high[:F])
    o = OracleNs(S, policy=2, reduce_mode=REDUCE_TREE32 if spw == 2 else REDUCE_TREE, fs=fs)
    ol, oh = o.run_bands(low, high)
    nbad = 0
    for rep in range(reps):
        g = ns.NsBatch(S, fs=fs, policy=2, streams_per_wave=spw)
        gl, gh = g.analyze_process_bands(low, high)
        bad = np.argwhere(gh.view(np.uint32) != oh.view(np.uint32))
        if len(bad):
            nbad += 1
            b0 = tuple(bad[0])
            print(fs, nh, spw, 'rep', rep, 'mismatches', len(bad), 'frames', np.unique(bad[:, 0])[:10], 'streams', np.unique(bad[:, 2]), 'first', gh[b0], oh[b0], 'ratio', oh[b0] / gh[b0], 'low ok', np.array_equal(gl.view(np.uint32), ol.view(np.uint32)))
    print(fs, nh, spw, 'bad reps', nbad, 'of', reps, os.environ.get('HIP_LAUNCH_BLOCKING'))
