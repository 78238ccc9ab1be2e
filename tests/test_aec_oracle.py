"""CPU suite: the AEC oracle.

oracle/aec_oracle.c is this repository's restatement of the reference AEC; it is PINNED here
against (a) the reference build oracle/_ref/libaec_ref.so run live (when present: the build
container), bit for bit on outputs, float state and control-plane integers, and (b) the committed
golden vectors tests/golden/aec_golden.npz (outputs of that same reference build), which travel."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest

from audiosignalprocess_amd._abi import aec_state_arrays
from audiosignalprocess_amd.synth import aec_frames
from tests import oracle_lib

needs_ref = pytest.mark.skipif(not oracle_lib.have_aec_ref(), reason="oracle/_ref/libaec_ref.so not built here")

TABLES = [(0, "rdft_w", 64), (1, "rdft_wk3ri_first", 16), (2, "rdft_wk3ri_second", 16),
          (3, "WebRtcAec_sqrtHanning", 65), (4, "WebRtcAec_weightCurve", 65),
          (5, "WebRtcAec_overDriveCurve", 65)]


@pytest.fixture(scope="module")
def aec_golden():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return dict(np.load(os.path.join(root, "tests", "golden", "aec_golden.npz")))


@pytest.fixture(scope="module")
def aec_ext_golden():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "aec_ext_golden.npz")))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


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
        assert np.array_equal(_bits(out), _bits(aec_golden["out_f32"][:, s]))


def test_oracle_reproduces_golden_bitwise(aec_golden):
    """The restatement against the reference's own outputs (travels to the GPU box)."""
    far, near = aec_golden["far_i16"].astype(np.float32), aec_golden["near_i16"].astype(np.float32)
    for s in range(far.shape[1]):
        out = oracle_lib.OracleAec().run(far[:, s], near[:, s])
        assert np.array_equal(_bits(out), _bits(aec_golden["out_f32"][:, s])), s


def test_oracle_tables_known_values():
    """Spot values of the generated tables against the decimal text of aec_rdft.c:32-61 and
    aec_core.c:53-98 (data, quoted to the printed precision)."""
    w = oracle_lib.aec_oracle_table(0, 64)
    assert w[0] == 1.0 and w[1] == 0.0
    assert abs(w[2] - 0.7071067691) < 1e-9 and abs(w[4] - 0.9238795638) < 1e-9
    assert abs(w[32] - 0.7071067691) < 1e-9 and abs(w[33] - 0.4993977249) < 1e-9
    assert abs(w[63] - 0.0245338380) < 1e-9
    h = oracle_lib.aec_oracle_table(3, 65)
    assert h[0] == 0.0 and h[64] == 1.0 and h[1] == np.float32(0.02454122852291)
    assert oracle_lib.aec_oracle_table(4, 65)[2] == np.float32(0.1378)
    assert oracle_lib.aec_oracle_table(5, 65)[1] == np.float32(1.1250)


@needs_ref
def test_oracle_tables_equal_reference_symbols():
    ref = oracle_lib.RefAec()
    for which, name, n in TABLES:
        assert np.array_equal(_bits(oracle_lib.aec_oracle_table(which, n)), _bits(ref.table(name, n))), name


def _fft_inputs():
    rng = np.random.default_rng(3)
    rows = [np.sin(np.arange(128, dtype=np.float32)),          # unittest_real_fft.cpp:29-31 style input
            np.eye(1, 128, 0, dtype=np.float32)[0], np.ones(128, np.float32),
            np.cos(np.pi * np.arange(128)).astype(np.float32)]
    rows += list((rng.standard_normal((12, 128)) * 3000).astype(np.float32))
    return np.stack(rows)


@needs_ref
def test_oracle_rdft128_equals_reference():
    ref = oracle_lib.RefAec()   # Create() installs the plain-C function table (aec_rdft_init)
    x = _fft_inputs()
    for isgn in (1, -1):
        assert np.array_equal(_bits(oracle_lib.aec_oracle_rdft128(x, isgn)), _bits(ref.rdft128(x, isgn)))


def test_oracle_rdft128_against_numpy():
    x = _fft_inputs()
    X = oracle_lib.aec_oracle_rdft128(x, 1).astype(np.float64)
    want = np.fft.rfft(x.astype(np.float64), axis=1)
    scale = np.abs(want).max(axis=1)
    # Ooura packing: a[0] = R0, a[1] = R64, a[2k] = Re, a[2k+1] = -Im of the textbook transform
    assert np.abs(X[:, 0] - want[:, 0].real).max() <= 1e-5 * scale.max()
    assert np.abs(X[:, 1] - want[:, 64].real).max() <= 1e-5 * scale.max()
    assert (np.abs(X[:, 2::2] - want[:, 1:64].real).max(axis=1) <= 1e-6 * scale).all()
    assert (np.abs(X[:, 3::2] + want[:, 1:64].imag).max(axis=1) <= 1e-6 * scale).all()
    # inverse(forward(x)) * 2/128 == x
    back = oracle_lib.aec_oracle_rdft128(oracle_lib.aec_oracle_rdft128(x, 1), -1) * np.float32(2.0 / 128)
    assert np.abs(back - x).max() <= 2e-6 * np.abs(x).max()


def _compare_states(a, b):
    sa, ca = a.export()
    sb, cb = b.export()
    da, db = aec_state_arrays(sa), aec_state_arrays(sb)
    for k in da:
        if isinstance(da[k], np.ndarray):
            assert np.array_equal(_bits(da[k]), _bits(db[k])), k
        else:
            assert da[k] == db[k], k
    for k in ("startup_phase", "checkBuffSize", "bufSizeStart", "knownDelay", "filtDelay",
              "timeForDelayChange", "lastDelayDiff", "counter", "sum", "firstVal", "checkBufSizeCtr",
              "system_delay", "core_knownDelay"):
        assert getattr(ca, k) == getattr(cb, k), k
    return ca, cb


@needs_ref
@pytest.mark.parametrize("fs,n,delay,nlp", [(16000, 160, 0, 1), (16000, 80, 40, 2), (8000, 80, 0, 0),
                                             (16000, 160, 120, 1)])
def test_oracle_equals_reference_live(fs, n, delay, nlp):
    """Free-running, frame by frame: outputs, float state and control-plane integers bit-equal,
    through the start-up phase, delay changes (incl. out-of-range reports) and both rates."""
    F = 420
    far, near = aec_frames(3, F * 160 // n if n == 80 else F)
    far = far.reshape(-1, 3, 160)[:, 2].reshape(-1, n)[:F]
    near = near.reshape(-1, 3, 160)[:, 2].reshape(-1, n)[:F]
    ref, ora = oracle_lib.RefAec(fs), oracle_lib.OracleAec(fs)
    assert ref.set_nlp(nlp) == 0 and ora.set_nlp(nlp) == 0
    for f in range(F):
        d = delay
        if f in (200, 201):
            d = 700        # > kMaxTrustedDelayMs: warning -1, still processes (echo_cancellation.c:367-375)
        if f == 250:
            d = -5         # negative: clamped to 0 with a warning
        if 300 <= f < 360:
            d = delay + 60  # a sustained change drives EstBufDelayNormal's knownDelay update
        o_ref, rc_ref = ref.frame(far[f], near[f], d)
        o_ora, rc_ora = ora.frame(far[f], near[f], d)
        assert rc_ref == rc_ora, f
        assert np.array_equal(_bits(o_ref), _bits(o_ora)), f
        if f % 35 == 0 or f == F - 1:
            ca, cb = _compare_states(ref, ora)
            # the reference's ring positions are private; compare the readable counts
            wrap = lambda r, w, wr, n_: (w - r) if wr == 0 else (n_ - r + w)
            assert ca.far_read == wrap(cb.far_read, cb.far_write, cb.far_wrap, 250)
            assert ca.pre_read == wrap(cb.pre_read, cb.pre_write, cb.pre_wrap, 448)
            assert ca.near_read == wrap(cb.near_read, cb.near_write, cb.near_wrap, 144)
            assert ca.out_read == wrap(cb.out_read, cb.out_write, cb.out_wrap, 144)


@needs_ref
@pytest.mark.parametrize("fs,n,nlp", [(16000, 160, 1), (16000, 80, 2), (8000, 80, 0)])
def test_oracle_extended_filter_equals_reference_live(fs, n, nlp):
    """WebRtcAec_enable_delay_correction(core, 1) -- the extended filter: 32 partitions, kExtendedMu /
    kExtendedErrorThreshold, the extended smoothing coefficients and minimum overdrive, no filter reset,
    ProcessExtended / EstBufDelayExtended (aec_core.c:172-174, 337-338, 383, 872-873, 1876-1881,
    echo_cancellation.c:744-814, 869-922): outputs, float state over all 32 partitions and the
    control-plane integers bit-equal frame by frame, through delay changes and out-of-range reports."""
    F = 460
    far, near = aec_frames(3, F * 160 // n if n == 80 else F)
    far = far.reshape(-1, 3, 160)[:, 1].reshape(-1, n)[:F]
    near = near.reshape(-1, 3, 160)[:, 1].reshape(-1, n)[:F]
    ref, ora = oracle_lib.RefAec(fs), oracle_lib.OracleAec(fs)
    assert ref.set_nlp(nlp) == 0 and ora.set_nlp(nlp) == 0
    ref.enable_delay_correction(1)
    ora.enable_delay_correction(1)
    for f in range(F):
        d = 30
        if f in (150, 151):
            d = 700         # >= kMaxTrustedDelayMs: replaced by kFixedDelayMs (ec:766-768)
        if f == 200:
            d = -5
        if 250 <= f < 330:
            d = 180         # a sustained change: EstBufDelayExtended's knownDelay update
        if f >= 400:
            d = 5           # below kMinTrustedDelayMs
        o_ref, rc_ref = ref.frame(far[f], near[f], d)
        o_ora, rc_ora = ora.frame(far[f], near[f], d)
        assert rc_ref == rc_ora, f
        assert np.array_equal(_bits(o_ref), _bits(o_ora)), f
        if f % 40 == 0 or f == F - 1:
            _compare_states(ref, ora)
    # switching the mode off again returns to the 12-partition filter (aec_core.c:1878)
    ref.enable_delay_correction(0)
    ora.enable_delay_correction(0)
    for f in range(40):
        o_ref, _ = ref.frame(far[f], near[f], 30)
        o_ora, _ = ora.frame(far[f], near[f], 30)
        assert np.array_equal(_bits(o_ref), _bits(o_ora)), f
    _compare_states(ref, ora)


def test_oracle_extended_filter_reproduces_golden(aec_ext_golden):
    """The reference's own extended-filter outputs (committed fixture) from the restatement, bit for bit."""
    far, near = aec_ext_golden["far_i16"].astype(np.float32), aec_ext_golden["near_i16"].astype(np.float32)
    for s in range(far.shape[1]):
        o = oracle_lib.OracleAec(16000)
        o.enable_delay_correction(1)
        out = o.run(far[:, s], near[:, s], int(aec_ext_golden["delay_ms"]))
        assert np.array_equal(_bits(out), _bits(aec_ext_golden["out_f32"][:, s])), s


def test_oracle_error_behaviour():
    """Return codes and lastError of echo_cancellation.c:196-215,278-300,341-375,410-438."""
    o = oracle_lib.OracleAec(16000)
    z = np.zeros(160, np.float32)
    out, rc = o.frame(z[:100], z[:100])
    assert rc == -1 and o.error_code() == 12004
    assert oracle_lib.OracleAec(44100).init_rc == -1
    assert oracle_lib.OracleAec(16000, sc_fs=0).init_rc == -1
    assert o.set_nlp(3) == -1 and o.error_code() == 12004
    assert o.set_nlp(1, skew=2) == -1 and o.error_code() == 12004
    assert o.set_nlp(1, delay_logging=2) == -1 and o.error_code() == 12004
    assert o.set_nlp(1, skew=1, delay_logging=1) == 0
    assert o.set_nlp(1, metrics=2) == -1 and o.error_code() == 12004
    assert o.set_nlp(1, metrics=1) == 0
    assert o.set_nlp(2) == 0


def test_oracle_metrics_reproduce_golden(aec_golden):
    """metricsMode = kAecTrue: UpdateLevel / UpdateMetrics / WebRtcAec_GetMetrics (aec_core.c:585-770,
    echo_cancellation.c:456-548) against the reference's own values in the fixture, bit for bit."""
    far, near = aec_golden["far_i16"].astype(np.float32), aec_golden["near_i16"].astype(np.float32)
    F, S = far.shape[:2]
    for s in range(S):
        o = oracle_lib.OracleAec(16000)
        assert o.set_nlp(1, metrics=1) == 0
        for f in range(F):
            out, rc = o.frame(far[f, s], near[f, s])
            assert rc == 0
            if f + 1 in (F // 2, F):
                k = 0 if f + 1 == F // 2 else 1
                assert np.array_equal(o.metrics_state().to_array(), aec_golden["met_state_u32"][k, s])
                assert o.get_metrics().to_tuple() == tuple(aec_golden["met_levels_i32"][k, s])
        assert np.array_equal(_bits(out), _bits(aec_golden["out_f32"][F - 1, s]))   # metrics do not touch the audio
    assert aec_golden["met_levels_i32"][1, :, 8].min() > 5                        # the fixture's ERLE is a real one


@needs_ref
def test_oracle_metrics_equal_reference_live():
    """Every frame of a bursty far end with a synthetic echo path: the metrics image never differs."""
    rng = np.random.default_rng(5)
    F = 900
    far = (rng.standard_normal((F, 160)) * 3000).astype(np.float32)
    far *= np.repeat((rng.random(F // 20) > 0.3).astype(np.float32), 20)[:, None] * 0.98 + 0.02
    h = (rng.standard_normal(200) * np.exp(-np.arange(200) / 40)).astype(np.float32) * 0.3
    near = np.convolve(far.reshape(-1), h)[:F * 160].astype(np.float32)
    near = (near + rng.standard_normal(F * 160).astype(np.float32) * 30).reshape(F, 160)
    r, o = oracle_lib.RefAec(16000), oracle_lib.OracleAec(16000)
    assert r.set_nlp(2, metrics=1) == 0 and o.set_nlp(2, metrics=1) == 0
    for f in range(F):
        a, _ = r.frame(far[f], near[f], 40)
        b, _ = o.frame(far[f], near[f], 40)
        assert np.array_equal(_bits(a), _bits(b))
        assert np.array_equal(r.metrics_state().to_array(), o.metrics_state().to_array()), f
    assert r.get_metrics().to_tuple() == o.get_metrics().to_tuple()
    assert o.metrics_state().erle.counter > 3
    # a second set_config restarts the statistics (aec_core.c:1858-1861)
    assert r.set_nlp(2, metrics=1) == 0 and o.set_nlp(2, metrics=1) == 0
    assert np.array_equal(r.metrics_state().to_array(), o.metrics_state().to_array())
    assert o.metrics_state().erle.counter == 0


def test_oracle_threaded_runner_matches_single(aec_golden):
    far, near = aec_golden["far_i16"][:120].astype(np.float32), aec_golden["near_i16"][:120].astype(np.float32)
    out = oracle_lib.aec_oracle_run_mt(far, near, threads=2)
    assert np.array_equal(_bits(out), _bits(aec_golden["out_f32"][:120]))


def test_host_control_plane_matches_oracle():
    """The HIP library's host-side control plane (aec_api.hip: start-up phase, system-delay
    bookkeeping, ring read/write positions, block scheduling) run WITHOUT a device
    (AspAecBatch_CreateControlOnly) against the oracle's -- which the tests above pin to the
    reference -- after every call of a sequence with delay jumps, both call sizes and both rates."""
    import ctypes as C

    from audiosignalprocess_amd import aec as aec_mod
    from audiosignalprocess_amd._abi import AspAecControl

    lib = aec_mod._lib()
    lib.AspAecBatch_CreateControlOnly.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    from audiosignalprocess_amd._abi import AecConfig
    for fs, n, ext, skw in [(16000, 160, 0, 0), (16000, 80, 0, 0), (8000, 80, 0, 0), (16000, 160, 1, 0), (8000, 80, 1, 0),
                            (16000, 160, 0, 17), (8000, 80, 0, -11)]:
        h = C.c_void_p()
        assert lib.AspAecBatch_CreateControlOnly(C.byref(h), 7) == 0
        assert lib.AspAecBatch_Init(h, fs, 48000) == 0
        ora = oracle_lib.OracleAec(fs)
        if skw:     # skewMode: GetSkew / EstimateSkew and the resampler's sample counts (system_delay follows them)
            assert lib.AspAecBatch_set_config(h, AecConfig(1, 1, 0, 0)) == 0 and ora.set_nlp(1, skew=1) == 0
        if ext:     # ProcessExtended / EstBufDelayExtended (echo_cancellation.c:744-814, 869-922)
            assert lib.AspAecBatch_enable_delay_correction(h, 1) == 0
            assert lib.AspAecBatch_delay_correction_enabled(h) == 1
            ora.enable_delay_correction(1)
        dummy = np.zeros(7 * 160, np.float32)
        z = np.zeros(n, np.float32)
        for f in range(900):
            d = 0
            if f in (120, 121):
                d = 700
            if f == 300:
                d = -3
            if 400 <= f < 520:
                d = 90
            if 600 <= f < 640:
                d = 20
            sk = (skw + (f * 7) % 5 - 2) if f % 89 else 4000
            rc_o = ora.frame_skew(z, z, d, sk)[1]
            rc_b = lib.AspAecBatch_BufferFarend(h, dummy.ctypes.data, n, 1)
            rc_b |= lib.AspAecBatch_Process(h, dummy.ctypes.data, dummy.ctypes.data, n, d, sk, 1)
            assert (rc_b != 0) == (rc_o != 0), (fs, n, ext, skw, f)
            cb = AspAecControl()
            assert lib.AspAecBatch_GetControl(h, C.byref(cb)) == 0
            _, co = ora.export()
            for name, _t in cb._fields_:
                assert getattr(cb, name) == getattr(co, name), (fs, n, ext, skw, f, name)
        assert lib.AspAecBatch_get_error_code(h) == ora.error_code()
        # data-touching entry points refuse a control-only handle
        lib.AspAecBatch_Synchronize.argtypes = [C.c_void_p]
        assert lib.AspAecBatch_Synchronize(h) != 0
        assert lib.AspAecBatch_Free(h) == 0


def _aec_band_frames(F, seed=5):
    """Far / near low band as the 16 kHz generator, plus a high band that carries echo-like and
    independent content (int16-range floats)."""
    far, near = aec_frames(2, F)
    rng = np.random.default_rng(seed)
    high = (0.3 * near[:, 1] + rng.standard_normal((F, 160)).astype(np.float32) * 200).astype(np.float32)
    high[120, :4] = 40000.0    # saturation of the scaled high band
    return far[:, 0].copy(), near[:, 0].copy(), high


@needs_ref
def test_oracle_two_bands_equals_reference_live():
    """32 kHz: the high band (delay line, average NLP gain of the upper half of the low band,
    H-band comfort noise: aec_core.c:501-545, 1032-1067) bit-equal to the reference, with the low
    band, the float state and the control plane, frame by frame."""
    F = 420
    far, nl, nh = _aec_band_frames(F)
    ref, ora = oracle_lib.RefAec(32000), oracle_lib.OracleAec(32000)
    rhs = np.empty_like(nh)
    for f in range(F):
        d = 40 if 250 <= f < 330 else 0
        rl, rh, rc_r = ref.frame_bands(far[f], nl[f], nh[f], d)
        rhs[f] = rh
        ol, oh, rc_o = ora.frame_bands(far[f], nl[f], nh[f], d)
        assert rc_r == rc_o, f
        assert np.array_equal(_bits(rl), _bits(ol)), f
        assert np.array_equal(_bits(rh), _bits(oh)), f
        if f % 60 == 0 or f == F - 1:
            _compare_states(ref, ora)
            st, _ = ora.export()
            # the reference keeps [previous | current] 64-sample halves; the first half is the carried one
            assert np.array_equal(_bits(ref.dbufh()[:64]), _bits(np.ctypeslib.as_array(st.dBufH)[:64])), f
    # the high band is attenuated while the far end is loud and the near end silent of speech
    assert np.abs(rhs[170:240]).mean() < 0.9 * np.abs(nh[170:240]).mean()
    assert oracle_lib.OracleAec(48000).init_rc == -1


def _delayed_near(S, F, lag_frames, stream=0):
    """far / near of the synthetic generator with the near end (echo included) `lag_frames` x 10 ms late:
    an echo path delay the 12-partition filter cannot span, which the delay estimator has to find."""
    far, near = aec_frames(S, F + lag_frames)
    far, near = far[:, stream], near[:, stream]
    return far[lag_frames:].copy(), near[:F].copy()


@needs_ref
@pytest.mark.parametrize("fs,n,ext", [(16000, 160, 0), (16000, 80, 0), (8000, 80, 0), (16000, 160, 1)])
def test_oracle_delay_logging_equals_reference_live(fs, n, ext):
    """set_config(delay_logging = kAecTrue): the binary-spectrum delay estimator (utility/delay_estimator.c,
    delay_estimator_wrapper.c float path, robust validation on) fed per block (aec_core.c:1191-1203), its whole
    state, the logging histogram and WebRtcAec_GetDelayMetrics (echo_cancellation.c:550-571, aec_core.c:1780-1836)
    equal to the reference frame by frame; the audio path is untouched by it."""
    F = 520 * 160 // n
    far, near = _delayed_near(2, 520, 6)
    far, near = far.reshape(-1, n)[:F], near.reshape(-1, n)[:F]
    ref, ora = oracle_lib.RefAec(fs), oracle_lib.OracleAec(fs)
    assert ref.delay_metrics()[0] == -1 and ora.delay_metrics()[0] == -1        # logging disabled
    assert ref.error_code() == 12001 and ora.error_code() == 12001
    assert ref.set_config(1, delay_logging=1) == 0 and ora.set_nlp(1, delay_logging=1) == 0
    if ext:
        ref.enable_delay_correction(1)
        ora.enable_delay_correction(1)
    assert ref.delay_metrics() == ora.delay_metrics() == (0, -1, -1)            # no values yet
    seen = set()
    for f in range(F):
        o_ref, rc_ref = ref.frame(far[f], near[f], 20)
        o_ora, rc_ora = ora.frame(far[f], near[f], 20)
        assert rc_ref == rc_ora and np.array_equal(_bits(o_ref), _bits(o_ora)), f
        if f % 20 == 19 or f == F - 1:
            assert ora.delay_state().diff(ref.delay_state()) == [], f
        if f % (130 * 160 // n) == 129:
            m_ref, m_ora = ref.delay_metrics(), ora.delay_metrics()
            assert m_ref == m_ora, f
            seen.add(m_ref)
    ca, _ = _compare_states(ref, ora)
    if ca.startup_phase == 0:   # (80-sample calls at 16 kHz never leave the reference's start-up phase: nBlocks10ms = 0)
        assert any(m[1] >= 0 for m in seen), seen       # the estimator did find the echo path's delay


@needs_ref
@pytest.mark.parametrize("fs,n,lag,ext", [(16000, 160, 9, 0), (16000, 80, 9, 0), (8000, 80, 5, 0), (16000, 160, 14, 1),
                                          (16000, 160, 0, 0)])
def test_oracle_delay_agnostic_equals_reference_live(fs, n, lag, ext):
    """WebRtcAec_enable_reported_delay(core, 0) with delay logging on -- the delay-agnostic mode:
    SignalBasedDelayCorrection (aec_core.c:797-850) moves the far-end read pointer by the estimated delay instead
    of the reported one (:1719-1732), the estimator is soft-reset by the move (delay_estimator.c:309-339, 500-511),
    EstBufDelay is skipped (echo_cancellation.c:725-727, 796-798).  Outputs, float state, control integers and
    the estimator's state equal to the reference frame by frame, through at least one correction."""
    F = 700 * 160 // n
    far, near = _delayed_near(2, 700, lag, stream=1)
    far, near = far.reshape(-1, n)[:F], near.reshape(-1, n)[:F]
    ref, ora = oracle_lib.RefAec(fs), oracle_lib.OracleAec(fs)
    assert ref.set_config(1, delay_logging=1) == 0 and ora.set_nlp(1, delay_logging=1) == 0
    ref.enable_reported_delay(0)
    ora.enable_reported_delay(0)
    if ext:
        ref.enable_delay_correction(1)
        ora.enable_delay_correction(1)
    for f in range(F):
        d = 700 if f in (300, 301) else 40
        o_ref, rc_ref = ref.frame(far[f], near[f], d)
        o_ora, rc_ora = ora.frame(far[f], near[f], d)
        assert rc_ref == rc_ora, f
        assert np.array_equal(_bits(o_ref), _bits(o_ora)), f
        if f % 25 == 24 or f == F - 1:
            assert ora.delay_state().diff(ref.delay_state()) == [], f
            _compare_states(ref, ora)
    d = ora.delay_state()
    print("delay-agnostic fs=%d n=%d lag=%d: corrections %d, last_delay %d, lookahead %d"
          % (fs, n, lag, d.delay_correction_count, d.last_delay, d.lookahead))
    if lag and ora.export()[1].startup_phase == 0:
        assert d.delay_correction_count >= 1


@needs_ref
@pytest.mark.parametrize("fs,n,sc_skew", [(16000, 160, 12), (16000, 80, -9), (8000, 80, 30)])
def test_oracle_skew_mode_equals_reference_live(fs, n, sc_skew):
    """set_config(skewMode = kAecTrue): the skew estimate from the reported per-call sample-count differences
    (WebRtcAec_GetSkew / EstimateSkew, aec_resampler.c:125-217; echo_cancellation.c:614-645) and the linear
    resampling of the far end (WebRtcAec_ResampleLinear, aec_resampler.c:74-123; echo_cancellation.c:304-313,
    831-833): outputs, state and control plane equal to the reference frame by frame, across the estimate
    (call 425) and the resampled frames after it."""
    F = 560
    far, near = aec_frames(2, F * 160 // n if n == 80 else F)
    far = far.reshape(-1, 2, 160)[:, 0].reshape(-1, n)[:F]
    near = near.reshape(-1, 2, 160)[:, 0].reshape(-1, n)[:F]
    ref, ora = oracle_lib.RefAec(fs), oracle_lib.OracleAec(fs)
    assert ref.set_config(1, skew=1) == 0 and ora.set_nlp(1, skew=1) == 0
    rng = np.random.default_rng(3)
    resampled = 0
    for f in range(F):
        sk = int(sc_skew + rng.integers(-2, 3))
        if f % 97 == 0:
            sk = 5000          # an outlier the estimator has to reject
        o_ref, rc_ref = ref.frame_skew(far[f], near[f], 30, sk)
        o_ora, rc_ora = ora.frame_skew(far[f], near[f], 30, sk)
        assert rc_ref == rc_ora, f
        assert np.array_equal(_bits(o_ref), _bits(o_ora)), f
        s_ref, s_ora = ref.skew_state(), ora.skew_state()
        assert np.float32(s_ref[0]).view(np.uint32) == np.float32(s_ora[0]).view(np.uint32) and s_ref[1] == s_ora[1], f
        resampled += s_ref[1]
        if f % 40 == 39 or f == F - 1:
            _compare_states(ref, ora)
    assert resampled > 50       # the far end was resampled for the rest of the run


def test_oracle_optional_modes_reproduce_golden():
    """tests/golden/aec_modes_golden.npz (written by the reference build): the delay-agnostic mode with its
    GetDelayMetrics values, and the skew mode, from the restatement bit for bit."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "aec_modes_golden.npz")))
    far, near = g["agn_far_i16"].astype(np.float32), g["agn_near_i16"].astype(np.float32)
    F, S = far.shape[:2]
    at = list(g["agn_metrics_at"])
    for s in range(S):
        o = oracle_lib.OracleAec(16000)
        assert o.set_nlp(1, delay_logging=1) == 0
        o.enable_reported_delay(0)
        for f in range(F):
            out, rc = o.frame(far[f, s], near[f, s], 40)
            assert rc == 0 and np.array_equal(_bits(out), _bits(g["agn_out_f32"][f, s])), (s, f)
            if f in at:
                assert tuple(g["agn_metrics"][at.index(f), s]) == o.delay_metrics(), (s, f)
    far, near = g["skew_far_i16"].astype(np.float32), g["skew_near_i16"].astype(np.float32)
    for s in range(S):
        o = oracle_lib.OracleAec(16000)
        assert o.set_nlp(1, skew=1) == 0
        for f in range(F):
            out, _ = o.frame_skew(far[f, s], near[f, s], 30, int(g["skew_arg"][f]))
            assert np.array_equal(_bits(out), _bits(g["skew_out_f32"][f, s])), (s, f)


@needs_ref
def test_oracle_two_bands_optional_modes_equal_reference_live():
    """32 kHz (two bands) with the extended filter, delay logging and the delay-agnostic mode together: low and high
    band, float state, control plane and the estimator's state equal to the reference frame by frame."""
    F = 460
    far, nl, nh = _aec_band_frames(F + 7)
    far, nl, nh = far[7:], nl[:F], nh[:F]          # near end 70 ms late
    ref, ora = oracle_lib.RefAec(32000), oracle_lib.OracleAec(32000)
    assert ref.set_config(1, delay_logging=1) == 0 and ora.set_nlp(1, delay_logging=1) == 0
    for o in (ref, ora):
        o.enable_reported_delay(0)
        o.enable_delay_correction(1)
    for f in range(F):
        rl, rh, rc_r = ref.frame_bands(far[f], nl[f], nh[f], 30)
        ol, oh, rc_o = ora.frame_bands(far[f], nl[f], nh[f], 30)
        assert rc_r == rc_o, f
        assert np.array_equal(_bits(rl), _bits(ol)) and np.array_equal(_bits(rh), _bits(oh)), f
        if f % 60 == 59 or f == F - 1:
            _compare_states(ref, ora)
            assert ora.delay_state().diff(ref.delay_state()) == [], f
    assert ora.delay_state().delay_correction_count >= 1
