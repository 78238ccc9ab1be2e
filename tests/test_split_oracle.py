"""CPU suite: the SplittingFilter oracle (two-band split at 32 kHz, three-band split at 48 kHz:
splitting_filter.cc:28-170).  oracle/split_oracle.c composes the QMF and sinc-resampler oracles the
way the reference composes its own; pinned bit for bit against the reference's C++ compiled in
place (oracle/_ref/libsplit_ref.so) and against tests/golden/split_golden.npz, which travels."""
import os

import numpy as np
import pytest

from tests import oracle_lib
from tests.test_sinc_oracle import sinc_inputs

needs_ref = pytest.mark.skipif(not oracle_lib.have_split_ref(), reason="oracle/_ref/libsplit_ref.so not built here")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@needs_ref
@pytest.mark.parametrize("nb", [2, 3])
def test_oracle_equals_reference(nb):
    x = sinc_inputs(80, 160 * nb, seed=23)
    ref, ora = oracle_lib.RefSplit(nb), oracle_lib.OracleSplit(nb)
    for f in range(x.shape[0]):
        br, bo = ref.analysis(x[f]), ora.analysis(x[f])
        assert np.array_equal(br, bo), f
        proc = bo.copy()
        proc[1:] = (proc[1:].astype(np.int32) * 5 // 8).astype(np.int16)   # as if a suppressor scaled the high bands
        assert np.array_equal(ref.synthesis(proc), ora.synthesis(proc)), f


def test_oracle_reproduces_golden():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "split_golden.npz")))
    ora = oracle_lib.OracleSplit(3)
    for f in range(g["x48"].shape[0]):
        bands = ora.analysis(g["x48"][f])
        assert np.array_equal(bands, g["bands"][f]), f
        assert np.array_equal(ora.synthesis(bands), g["merged"][f]), f
