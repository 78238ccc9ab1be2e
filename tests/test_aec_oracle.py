"""CPU suite: the AEC reference oracle (row c of SURVEY section 8) is pinned for the next round.

No HIP AEC path exists yet; what is checked here is that the reference build under oracle/_ref
reproduces the committed golden vectors bit for bit (so the fixture and its generator stay in
sync) and that the fixture shows the behaviour the kernels will have to match."""
import numpy as np
import pytest

from tests import oracle_lib

needs_ref = pytest.mark.skipif(not oracle_lib.have_aec_ref(), reason="oracle/_ref/libaec_ref.so not built here")


@pytest.fixture(scope="module")
def aec_golden():
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return dict(np.load(os.path.join(root, "tests", "golden", "aec_golden.npz")))


def test_golden_shapes_and_echo_suppression(aec_golden):
    far, near, out = aec_golden["far_i16"], aec_golden["near_i16"], aec_golden["out_f32"]
    assert far.shape == near.shape == out.shape and out.shape[2] == 160
    assert np.isfinite(out).all() and np.abs(out).max() <= 32768
    # loud far-end, no near-end talker (frames 150..299): the echo is cancelled by > 15 dB
    seg = slice(170, 290)
    erle = 10 * np.log10((near[seg].astype(np.float64) ** 2).mean() / (out[seg].astype(np.float64) ** 2).mean())
    assert erle > 15, erle
    # start-up: the first frames are passed through while the buffers fill (echo_cancellation.c:646-722)
    assert np.array_equal(out[0], near[0].astype(np.float32))


@needs_ref
def test_reference_reproduces_golden_bitwise(aec_golden):
    far, near = aec_golden["far_i16"].astype(np.float32), aec_golden["near_i16"].astype(np.float32)
    for s in range(far.shape[1]):
        out = oracle_lib.RefAec().run(far[:, s], near[:, s])
        assert np.array_equal(out.view(np.uint32), aec_golden["out_f32"][:, s].view(np.uint32))
