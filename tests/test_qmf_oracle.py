"""CPU suite: the two-band QMF oracle (the 32 kHz band split).

oracle/qmf_oracle.c restates WebRtcSpl_AnalysisQMF / WebRtcSpl_SynthesisQMF; it is pinned bit for
bit against the reference file compiled in place (oracle/_ref/libspl_ref.so, build container) and
against tests/golden/qmf_golden.npz (outputs of that build), which travels."""
import os

import numpy as np
import pytest

from tests import oracle_lib

needs_ref = pytest.mark.skipif(not oracle_lib.have_spl_ref(), reason="oracle/_ref/libspl_ref.so not built here")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def qmf_inputs(frames=40, n=320, seed=11):
    """Speech-like, full-scale square (saturation paths), impulses and silence, frame by frame."""
    rng = np.random.default_rng(seed)
    t = np.arange(frames * n)
    x = (6000 * np.sin(0.05 * t) + 3000 * np.sin(0.9 * t) + rng.normal(0, 800, t.size))
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(frames, n)
    x[5] = 32767
    x[6] = -32768
    x[7, ::2] = 32767
    x[7, 1::2] = -32768
    x[8] = 0
    x[9] = 0
    x[9, 0] = 32767
    x[10] = rng.integers(-32768, 32768, n).astype(np.int16)
    return x


@needs_ref
@pytest.mark.parametrize("n", [320, 160, 640])
def test_oracle_equals_reference(n):
    x = qmf_inputs(n=n)
    ref, ora = oracle_lib.RefQmf(), oracle_lib.OracleQmf()
    for f in range(x.shape[0]):
        lr, hr = ref.analysis(x[f])
        lo, ho = ora.analysis(x[f])
        assert np.array_equal(lr, lo) and np.array_equal(hr, ho), f
        # synthesis fed with processed-looking bands (scaled high band), states carried
        hb = (hr.astype(np.int32) * 3 // 4).astype(np.int16)
        assert np.array_equal(ref.synthesis(lr, hb), ora.synthesis(lo, hb)), f
        assert np.array_equal(ref.state(), ora.state()), f


def test_oracle_reproduces_golden():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "qmf_golden.npz")))
    ora = oracle_lib.OracleQmf()
    for f in range(g["x"].shape[0]):
        lo, hi = ora.analysis(g["x"][f])
        assert np.array_equal(lo, g["low"][f]) and np.array_equal(hi, g["high"][f]), f
        assert np.array_equal(ora.synthesis(lo, hi), g["merged"][f]), f
    assert np.array_equal(ora.state(), g["state"])


def test_split_merge_is_power_complementary():
    """QMF property: merge(split(x)) keeps the signal's energy (all-pass pair: power complementary,
    phase distorted) and stays strongly correlated with x at the bank's delay of 4 samples."""
    x = qmf_inputs(frames=30)
    x[5:11] = x[11:17]          # no saturating frames here
    ora = oracle_lib.OracleQmf()
    y = np.concatenate([ora.synthesis(*ora.analysis(fr)) for fr in x]).astype(np.float64)
    xs = x.reshape(-1).astype(np.float64)
    assert abs(np.mean(y[2000:] ** 2) / np.mean(xs[2000:] ** 2) - 1.0) < 0.01
    d = 4
    c = np.corrcoef(y[d + 2000:], xs[2000:xs.size - d])[0, 1]
    assert c > 0.95, c
