"""The reference's own test of the three-band splitting filter (modules/audio_processing/splitting_filter_unittest.cc:
30-104, `SplittingFilterTest.SplitsIntoThreeBandsAndReconstructs`; gtest is not vendored, so it cannot be built),
restated over the oracle, the reference compiled in place (where it exists) and -- marked gpu -- the batched HIP path:
eight 10 ms chunks at 48 kHz carrying every combination of a 1 kHz, a 12 kHz and an 18 kHz sine of amplitude 8192; a
band's energy is above amplitude^2 / 4 exactly when its sine is present, and the recombined signal correlates with the
input (best delay, as the reference searches it) above amplitude^2 / 4 whenever any sine is present."""
import numpy as np
import pytest

from tests import oracle_lib

K_SAMPLE_RATE_HZ, K_NUM_BANDS, K_CHUNKS = 48000, 3, 8
K_FREQUENCIES_HZ = (1000, 12000, 18000)
K_AMPLITUDE = 8192.0
N48, N16 = 480, 160


def _chunk(i):
    """chunk i of the reference's input (:48-60) as the int16 samples its IFChannelBuffer hands the filter"""
    k = np.arange(N48)
    x = np.zeros(N48, np.float32)
    present = [bool(i & (1 << j)) for j in range(K_NUM_BANDS)]
    for j in range(K_NUM_BANDS):
        amp = K_AMPLITUDE if present[j] else 0.0
        x += (amp * np.sin(2 * np.pi * K_FREQUENCIES_HZ[j] * (i * N48 + k) / K_SAMPLE_RATE_HZ)).astype(np.float32)
    # FloatS16ToS16 (common_audio/include/audio_util.h:41-49): round half away from zero, saturate
    xi = np.where(x > 0, np.floor(np.minimum(x, 32766.5) + 0.5), np.ceil(np.maximum(x, -32767.5) - 0.5))
    return xi.astype(np.int16), present


def _check(analysis, synthesis):
    for i in range(K_CHUNKS):
        x, present = _chunk(i)
        bands = analysis(x)
        for j in range(K_NUM_BANDS):
            energy = float((bands[j].astype(np.float32) ** 2).sum() / N16)
            if present[j]:
                assert energy > K_AMPLITUDE * K_AMPLITUDE / 4, (i, j, energy)
            else:
                assert energy < K_AMPLITUDE * K_AMPLITUDE / 4, (i, j, energy)
        out = synthesis(bands).astype(np.float32)
        xf = x.astype(np.float32)
        xcorr = 0.0
        for delay in range(N48):   # :80-91
            tmp = float((xf[delay:] * out[:N48 - delay]).sum() / N48)
            xcorr = max(xcorr, tmp)
        if any(present):
            assert xcorr > K_AMPLITUDE * K_AMPLITUDE / 4, (i, xcorr)


def test_oracle_splits_into_three_bands_and_reconstructs():
    o = oracle_lib.OracleSplit(3)
    _check(o.analysis, o.synthesis)


@pytest.mark.skipif(not oracle_lib.have_split_ref(), reason="oracle/_ref not built")
def test_reference_passes_its_own_test():
    r = oracle_lib.RefSplit(3)
    _check(r.analysis, r.synthesis)


@pytest.mark.gpu
def test_hip_splits_into_three_bands_and_reconstructs():
    from audiosignalprocess_amd import qmf

    g = qmf.SplitBatch(5, 3)       # five channels fed the same chunks: channel 3 is looked at
    _check(lambda x: g.analysis(np.repeat(x[None], 5, axis=0))[:, 3], lambda b: g.synthesis(np.repeat(b[:, None], 5, axis=1))[3])
    g.close()
