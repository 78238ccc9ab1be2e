"""GPU parity tests of the batched push sinc resampler (through the C-ABI, include/asp_resample.h):
bit-exact against oracle/sinc_oracle.c (pinned to the reference's C++ in tests/test_sinc_oracle.py)
and against the committed reference outputs."""
import os

import numpy as np
import pytest

from tests import oracle_lib
from tests.test_sinc_oracle import sinc_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rs():
    from audiosignalprocess_amd import ns
    from audiosignalprocess_amd import resample as mod

    assert ns.device_count() >= 1, "GPU tests need a HIP device"
    return mod


@pytest.mark.parametrize("src,dst", [(480, 640), (640, 480)])
def test_batch_equals_oracle_bitwise(rs, src, dst):
    """19 channels x 120 frames (the 4/3 ratio accumulates in double over all of them), every
    channel on its own rotation of the frames; kernel table equal to the oracle's."""
    Cn, F = 19, 120
    x = sinc_inputs(F, src)
    g = rs.SincBatch(Cn, src, dst)
    oras = [oracle_lib.OracleSinc(src, dst) for _ in range(Cn)]
    assert np.array_equal(g.kernel_table().view(np.uint32), oras[0].kernel().view(np.uint32))
    for f in range(F):
        frame = np.stack([x[(f + 3 * c) % F] for c in range(Cn)])
        y = g.resample(frame)
        for c in range(0, Cn, 4):
            assert np.array_equal(y[c], oras[c].resample(frame[c])), (f, c)


def test_golden_reference_outputs_and_scale(rs):
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "sinc_golden.npz")))
    up, down = rs.SincBatch(1, 480, 640), rs.SincBatch(1, 640, 480)
    for f in range(gold["x48"].shape[0]):
        y = up.resample(gold["x48"][f][None])
        assert np.array_equal(y[0], gold["y64"][f]), f
        assert np.array_equal(down.resample(y)[0], gold["z48"][f]), f
    big = rs.SincBatch(4096, 480, 640)
    for f in range(3):
        y = big.resample(np.broadcast_to(gold["x48"][f], (4096, 480)))
        assert (y == y[0]).all() and np.array_equal(y[0], gold["y64"][f])
