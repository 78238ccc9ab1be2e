"""CPU suite for the BlockThresholding oracle (oracle/bt_oracle.c).

PARITY UNPINNED: the reference's kiss_fft cannot be compiled (its _kiss_fft_guts.h is missing)
and it ships no expected outputs, so the restatement is anchored on an independent FFT (numpy),
round-trip identities and the reference's own framing protocol / return codes."""
import numpy as np
import pytest

from audiosignalprocess_amd.synth import bt_samples
from tests.oracle_lib import OracleBt

NEED_MORE, CAN_OUTPUT, ERR_PARAMS = 0x10, 0x20, 0x02


@pytest.mark.parametrize("n", [256, 1024])
def test_kiss_fftr_matches_numpy_and_roundtrips(n):
    o = OracleBt(n)
    rng = np.random.default_rng(n)
    i = np.arange(n)
    for x in [np.sin(i).astype(np.float32),  # the input family of unittest_real_fft.cpp:29-31
              (i == 0).astype(np.float32), np.ones(n, np.float32),
              np.where(i % 2 == 0, 1, -1).astype(np.float32),
              rng.standard_normal(n).astype(np.float32) * 0.3]:
        f = o.kiss_fftr(x)
        X = np.fft.rfft(x.astype(np.float64))  # kiss forward = textbook sign (kiss_fftr.c:92-120)
        assert np.abs((f[0::2] + 1j * f[1::2]) - X).max() <= 1e-6 * max(1.0, np.abs(X).max())
        assert f[1] == 0 and f[n + 1] == 0
        back = o.kiss_fftri(f) / n  # unscaled inverse (audioDenoiseBlockTreshold.c:297)
        assert np.abs(back - x).max() <= 2e-6 * max(1.0, np.abs(x).max())


def test_hann_window_is_the_symmetric_form():
    for n in (256, 1024):
        h = OracleBt(n).hann()
        ref = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / (n - 1))).astype(np.float32)
        assert np.array_equal(h, ref) and h[0] == 0 and h[-1] == 0


@pytest.mark.parametrize("n", [256, 1024])
def test_hop_protocol_and_macroblock_equivalence(n):
    """denoise x7 -> NEED_MORE, 8th -> CAN_OUTPUT (.c:541-575); output lags input by n/2."""
    o1, o2 = OracleBt(n), OracleBt(n)
    x = bt_samples(1, 3 * o1.macro)[0]
    out = []
    for k, hop in enumerate(x.reshape(-1, o1.half)):
        rc = o1.denoise_float(hop)
        assert rc == (CAN_OUTPUT if k % 8 == 7 else NEED_MORE)
        if rc == CAN_OUTPUT:
            assert o1.output_float(o1.macro - 1)[0] == 0  # buffer too small (.c:595-597)
            got, y = o1.output_float()
            assert got == o1.macro
            out.append(y)
    assert o1.denoise_float(x[:o1.half - 1]) == ERR_PARAMS  # in_len must equal half_win (.c:544)
    assert np.array_equal(np.concatenate(out), o2.run(x))
    assert np.isfinite(out[-1]).all()


def test_denoising_effect_and_segmentation_range():
    o = OracleBt(1024)
    x = bt_samples(1, 4 * o.macro, stream0=3)[0]
    y, seg = o.macroblock(x[:o.macro], want_seg=True)
    assert seg.shape == (31, 2) and seg[:, 0].max() <= 2 and seg[:, 1].max() <= 4 and seg.min() >= 0
    y = np.concatenate([y, o.run(x[o.macro:])])
    # white noise of sigma ~0.046 is attenuated well below its input level in the quiet half
    assert np.sqrt((y[5000:15000] ** 2).mean()) < 0.6 * np.sqrt((x[5000:15000] ** 2).mean())


def test_flush_passes_partial_macroblock_unthresholded():
    o = OracleBt(256)
    x = bt_samples(1, 3 * 128, stream0=9)[0]
    for hop in x.reshape(3, 128):
        assert o.denoise_float(hop) == NEED_MORE
    assert o.flush_float(3 * 128 - 1)[0] == -1
    got, y = o.flush_float(3 * 128)
    assert got == 3 * 128
    # overlap-add of Hann-windowed frames, output delayed by half a window:
    # y[128 + j] = x[j] * (w[j] + w[j + 128])
    h = o.hann()
    expect = x[:128] * (h[:128] + h[128:])
    assert np.abs(y[128:256] - expect).max() < 1e-5


def test_s16_conversions():
    o = OracleBt(256)
    f = o.lib.bt_oracle_s16_to_float
    g = o.lib.bt_oracle_float_to_s16
    assert f(32767) == 1.0 and f(-32768) == -1.0 and f(0) == 0.0
    assert g(1.5) == 32767 and g(-1.5) == -32768 and g(0.5) == 16384 and g(-0.5) == -16384
