"""CPU suite: the push sinc resampler oracle (48 <-> 64 kHz legs of the three-band split,
splitting_filter.cc:91-170).  oracle/sinc_oracle.c is pinned bit for bit against the reference's
C++ sources compiled in place (oracle/_ref/libsinc_ref.so, the SSE convolution an x86-64 build of
the reference uses) and against tests/golden/sinc_golden.npz, which travels."""
import os

import numpy as np
import pytest

from tests import oracle_lib

needs_ref = pytest.mark.skipif(not oracle_lib.have_sinc_ref(), reason="oracle/_ref/libsinc_ref.so not built here")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sinc_inputs(frames, n, seed=17):
    rng = np.random.default_rng(seed)
    t = np.arange(frames * n)
    x = 7000 * np.sin(0.031 * t) + 2500 * np.sin(0.7 * t) + rng.normal(0, 600, t.size)
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(frames, n)
    x[4] = 32767
    x[5] = -32768
    x[6] = 0
    x[7] = 0
    x[7, 100] = 32767
    x[8] = rng.integers(-32768, 32768, n).astype(np.int16)
    return x


@needs_ref
@pytest.mark.parametrize("src,dst", [(480, 640), (640, 480)])
def test_oracle_equals_reference(src, dst):
    """200 frames: long enough for the 4/3 ratio's double accumulation to wander if it ever would."""
    x = sinc_inputs(200, src)
    ref, ora = oracle_lib.RefSinc(src, dst), oracle_lib.OracleSinc(src, dst)
    for f in range(x.shape[0]):
        assert np.array_equal(ref.resample(x[f]), ora.resample(x[f])), f


def test_oracle_reproduces_golden():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "sinc_golden.npz")))
    up, down = oracle_lib.OracleSinc(480, 640), oracle_lib.OracleSinc(640, 480)
    for f in range(g["x48"].shape[0]):
        y = up.resample(g["x48"][f])
        assert np.array_equal(y, g["y64"][f]), f
        assert np.array_equal(down.resample(y), g["z48"][f]), f


def test_up_down_roundtrip_is_close():
    """48 -> 64 -> 48 kHz reproduces a band-limited signal, delayed by a non-integer number of
    samples near 29, to better than -25 dB at the nearest integer lag."""
    t = np.arange(60 * 480)
    x = (8000 * np.sin(0.02 * t) + 3000 * np.sin(0.21 * t)).astype(np.int16).reshape(60, 480)
    up, down = oracle_lib.OracleSinc(480, 640), oracle_lib.OracleSinc(640, 480)
    z = np.concatenate([down.resample(up.resample(fr)) for fr in x]).astype(np.float64)
    xs = x.reshape(-1).astype(np.float64)
    best = min(range(0, 64), key=lambda d: np.mean((z[d + 4000:] - xs[4000:xs.size - d]) ** 2))
    err = np.mean((z[best + 4000:] - xs[4000:xs.size - best]) ** 2) / np.mean(xs[4000:] ** 2)
    assert 20 <= best <= 40 and err < 3e-3, (best, err)
