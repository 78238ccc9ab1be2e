"""GPU parity suite (-m gpu): the hand-written HIP path, called through the C-ABI
(libasp_amd.so), against the oracle and the reference's golden vectors.

Bars:
  * bit-exact vs the oracle in ASP_NS_REDUCE_TREE mode (same float operations in the same
    order, including the wave64 reduction association);
  * <= 1e-4 relative (per-stream L2) vs the reference's outputs (golden vectors; the only
    difference is the association of the ~10 cross-bin sums per frame), BASELINE.json tolerance.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from audiosignalprocess_amd.synth import ns_frames
from tests.conftest import (parity_note, CHAOS_CAP, check_free_running, free_running_report, rel_l2_per_stream, state_diff,
                            state_from_bytes)
from tests.oracle_lib import REDUCE_SEQ, REDUCE_TREE, REDUCE_TREE32, REDUCE_TREE64P, OracleNs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL_TOL = 1e-4  # BASELINE.json north_star: <= 1e-4 relative vs the CPU reference


@pytest.fixture(scope="module")
def ns():
    from audiosignalprocess_amd import ns as mod

    assert mod.device_count() >= 1, "GPU tests need a HIP device"
    return mod


def test_fft_bit_exact(ns, golden):
    assert np.array_equal(ns.rdft256(golden["fft_in"], 1), golden["fft_fwd"])
    assert np.array_equal(ns.rdft256(golden["fft_fwd"], -1), golden["fft_inv"])
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((1000, 256)) * rng.choice([1e-3, 1.0, 3e3, 1e6], size=(1000, 1))).astype(np.float32)
    o = OracleNs(1)
    f = ns.rdft256(x, 1)
    assert np.array_equal(f, o.rdft256(x, 1))
    assert np.array_equal(ns.rdft256(f, -1), o.rdft256(f, -1))


def test_fft_linearity_and_roundtrip_large(ns):
    """Size-independent properties at a batch far beyond what the CPU checker covers."""
    rng = np.random.default_rng(5)
    n = 65536
    a = rng.standard_normal((n, 256)).astype(np.float32)
    fa = ns.rdft256(a, 1)
    back = ns.rdft256(fa, -1) * np.float32(2.0 / 256)
    assert np.abs(back - a).max() <= 2e-6 * np.abs(a).max() * 8
    # Parseval: sum x^2 == (R0^2 + R128^2 + 2 sum |X_k|^2) / 256
    e_t = (a.astype(np.float64) ** 2).sum(axis=1)
    e_f = (fa[:, 0].astype(np.float64) ** 2 + fa[:, 1].astype(np.float64) ** 2
           + 2 * (fa[:, 2:].astype(np.float64) ** 2).sum(axis=1)) / 256
    assert np.abs(e_f / e_t - 1).max() < 1e-5


def test_fused_free_running_bit_exact_vs_tree_oracle(ns):
    S, F = 24, 1100  # crosses blockInd 50 / 200 and two 500-frame histogram windows
    x = ns_frames(S, F)
    g = ns.NsBatch(S, policy=1, kernel=1)
    y = g.analyze_process(x)
    o = OracleNs(S, policy=1, reduce_mode=REDUCE_TREE)
    yo = o.run(x, threads=8)
    assert np.isfinite(y).all()
    assert np.array_equal(y, yo)
    for s in range(0, S, 5):
        assert state_diff(g.export_state(s), o.export_state(s)) == {}
    assert np.array_equal(g.prior_speech_probability(),
                          np.array([o.export_state(s).priorSpeechProb for s in range(S)], np.float32))
    g.close()


def test_golden_reference_outputs_within_tolerance(ns, golden):
    x = golden["in_i16"].astype(np.float32)
    F, S, _ = x.shape
    g = ns.NsBatch(S, policy=1, kernel=1)
    y = g.analyze_process(x)
    rel = rel_l2_per_stream(y, golden["out_f32"])
    check_free_running(rel)
    # robust frame-level statistic
    num = np.sqrt(((y - golden["out_f32"]).astype(np.float64) ** 2).sum(axis=2))
    den = np.sqrt((golden["out_f32"].astype(np.float64) ** 2).sum(axis=2)) + 1e-9
    assert np.percentile(num / den, 95) <= REL_TOL
    g.close()


def test_golden_teacher_forced_single_step(ns, golden):
    """Inject each reference snapshot, advance one frame, compare output and every state array
    with the reference one frame later (snapshots k, k+1 exist for 49/50/51, 199..202, 499..501)."""
    x = golden["in_i16"].astype(np.float32)
    S = x.shape[1]
    frames = [int(f) for f in golden["snap_frames"]]
    checked = 0
    for k, f0 in enumerate(frames):
        if f0 >= x.shape[0]:
            continue
        g = ns.NsBatch(S, policy=1, kernel=1)
        for s in range(S):
            g.import_state(s, state_from_bytes(golden["snap_state"][k, s]))
        y = g.analyze_process(x[f0:f0 + 1])
        ref_y = golden["out_f32"][f0:f0 + 1]
        scale = np.abs(ref_y).max()
        # the start-up pink-noise fit (ns_core.c:1113-1133) subtracts nearly equal sums, so the
        # first 50 frames amplify the reduction-order difference (measured 3.5e-5 at frame 1,
        # <= 1.2e-7 after start-up); the bar stays inside BASELINE's 1e-4.
        tol = REL_TOL if f0 < 50 else 1e-5
        assert np.abs(y - ref_y).max() <= tol * scale, f0
        if f0 + 1 in frames:
            k1 = frames.index(f0 + 1)
            for s in range(S):
                bad = state_diff(g.export_state(s), state_from_bytes(golden["snap_state"][k1, s]))
                ref_d = state_from_bytes(golden["snap_state"][k1, s]).to_dict()
                for name, (_, maxabs) in bad.items():
                    assert maxabs <= tol * max(1.0, float(np.abs(ref_d[name]).max())), (f0, s, name, bad)
            checked += 1
        g.close()
    assert checked >= 6


def test_unfused_analyze_process_equals_fused(ns):
    """Separate Analyze / Process launches (general state representation) == fused launch,
    including the paired -> unpaired transition mid-stream."""
    S, F = 7, 260  # ragged: not a multiple of the 4 streams per workgroup
    x = ns_frames(S, F, stream0=11)
    g1 = ns.NsBatch(S, policy=2, kernel=1)
    y1 = g1.analyze_process(x)
    g2 = ns.NsBatch(S, policy=2, kernel=1)
    y2 = np.empty_like(x)
    y2[:100] = g2.analyze_process(x[:100])  # fused / paired
    for f in range(100, F):                  # then the reference's two-call protocol
        g2.analyze(x[f])
        y2[f] = g2.process(x[f])
    assert np.array_equal(y1, y2)
    y3 = g2.analyze_process(x[:5])           # fused entry point after un-pairing still works
    y4 = g1.analyze_process(x[:5])
    assert np.array_equal(y3, y4)
    for s in range(S):
        assert state_diff(g1.export_state(s), g2.export_state(s)) == {}
    g1.close()
    g2.close()


def test_analyze_and_process_on_different_frames(ns):
    """Analyze(a) / Process(b) with a != b (legal in the reference API) vs the oracle."""
    S, F = 3, 120
    a = ns_frames(S, F, stream0=40)
    b = ns_frames(S, F, stream0=50)
    g = ns.NsBatch(S, policy=1, kernel=1)
    o = OracleNs(S, policy=1, reduce_mode=REDUCE_TREE)
    for f in range(F):
        g.analyze(a[f])
        o.analyze(a[f])
        assert np.array_equal(g.process(b[f]), o.process(b[f])), f
    for s in range(S):
        assert state_diff(g.export_state(s), o.export_state(s)) == {}
    g.close()


def test_zero_and_silence_edge_cases(ns):
    S, F = 5, 60
    x = ns_frames(S, F, stream0=3)
    x[10:16, 1] = 0.0      # stream 1: silence long enough for energy == 0 frames
    x[:, 2] = 0.0          # stream 2: digital silence from the start
    x[30:, 3] = 32767.0    # stream 3: full-scale DC (saturation path)
    g = ns.NsBatch(S, policy=3, kernel=1)
    o = OracleNs(S, policy=3, reduce_mode=REDUCE_TREE)
    y, yo = g.analyze_process(x), o.run(x)
    assert np.isfinite(y).all()
    assert np.array_equal(y, yo)
    assert (y[:, 2] == 0).all()
    assert np.abs(y).max() <= 32768.0
    for s in range(S):
        assert state_diff(g.export_state(s), o.export_state(s)) == {}
    assert g.export_state(2).blockInd == -1
    g.close()


def test_all_policies_bit_exact(ns):
    for policy in range(4):
        S, F = 4, 230
        x = ns_frames(S, F, stream0=20 * policy)
        g = ns.NsBatch(S, policy=policy, kernel=1)
        o = OracleNs(S, policy=policy, reduce_mode=REDUCE_TREE)
        assert np.array_equal(g.analyze_process(x), o.run(x)), policy
        g.close()


def test_state_export_import_roundtrip(ns):
    """Checkpoint / restore: a restored batch continues bit-identically."""
    S, F = 4, 300
    x = ns_frames(S, F + 50, stream0=77)
    g = ns.NsBatch(S, policy=1, kernel=1)
    g.analyze_process(x[:F])
    saved = [g.export_state(s) for s in range(S)]
    y_a = g.analyze_process(x[F:])
    g2 = ns.NsBatch(S, policy=0, kernel=1)
    for s in range(S):
        g2.import_state(s, saved[s])
    y_b = g2.analyze_process(x[F:])
    assert np.array_equal(y_a, y_b)
    g.close()
    g2.close()


def test_config2_scale_4096_streams(ns):
    """BASELINE config[1] size: 4096 streams.  Spot-check streams against the oracle and check
    size-independent properties on all of them."""
    S, F = 4096, 60
    x = ns_frames(S, F)
    g = ns.NsBatch(S, policy=1, kernel=1)
    y = g.analyze_process(x)
    assert np.isfinite(y).all() and np.abs(y).max() <= 32768.0
    pick = [0, 1, 2, 3, 63, 64, 1023, 2048, 4093, 4094, 4095]
    o = OracleNs(len(pick), policy=1, reduce_mode=REDUCE_TREE)
    yo = o.run(np.ascontiguousarray(x[:, pick]))
    assert np.array_equal(y[:, pick], yo)
    # streams are independent: permuting the batch permutes the outputs
    perm = np.random.default_rng(0).permutation(S)
    g2 = ns.NsBatch(S, policy=1, kernel=1)
    y2 = g2.analyze_process(np.ascontiguousarray(x[:, perm]))
    assert np.array_equal(y2, y[:, perm])
    # a noise suppressor never amplifies the frame energy by much after start-up
    e_in = (x[50:].astype(np.float64) ** 2).sum(axis=(0, 2))
    e_out = (y[50:].astype(np.float64) ** 2).sum(axis=(0, 2))
    assert (e_out <= 1.05 * e_in).all()
    g.close()
    g2.close()


def test_device_pointer_path_in_place(ns):
    """ASP_MEM_DEVICE with in == out (the driver aliases them, test_ns_module.cpp:98-99)."""
    S, F = 16, 40
    x = ns_frames(S, F, stream0=5)
    buf = ns.DeviceBuffer(x.nbytes)
    buf.upload(x)
    g = ns.NsBatch(S, policy=1, kernel=1)
    g.analyze_process_device(buf.ptr, buf.ptr, F)
    g.synchronize()
    y = buf.download(x.shape)
    assert np.array_equal(y, OracleNs(S, policy=1, reduce_mode=REDUCE_TREE).run(x))
    g.close()
    buf.free()


def test_layer1_reference_api(ns, built_lib):
    """The reference's own per-stream symbols (noise_suppression.h:16-122) over ctypes."""
    lib = C.CDLL(built_lib)
    lib.WebRtcNs_prior_speech_probability.restype = C.c_float
    h = C.c_void_p()
    assert lib.WebRtcNs_prior_speech_probability(None) == -1.0
    assert lib.WebRtcNs_Create(C.byref(h)) == 0
    assert lib.WebRtcNs_prior_speech_probability(h) == -1.0  # not initialised yet
    assert lib.WebRtcNs_Init(h, 44100) == -1
    assert lib.WebRtcNs_Init(h, 16000) == 0
    assert lib.WebRtcNs_set_policy(h, 7) == -1
    assert lib.WebRtcNs_set_policy(h, 1) == 0
    x = ns_frames(1, 80, stream0=9)
    o = OracleNs(1, policy=1, reduce_mode=REDUCE_TREE)
    yo = o.run(x)
    fp = C.POINTER(C.c_float)
    for f in range(80):
        frame = np.ascontiguousarray(x[f, 0])
        ptr = frame.ctypes.data_as(fp)
        lib.WebRtcNs_Analyze(h, ptr)
        bands = (fp * 1)(ptr)
        lib.WebRtcNs_Process(h, bands, 1, bands)  # in place
        assert np.array_equal(frame, yo[f, 0]), f
    assert lib.WebRtcNs_prior_speech_probability(h) == np.float32(o.export_state(0).priorSpeechProb)
    assert lib.WebRtcNs_Free(h) == 0


def test_wav_driver_end_to_end(ns, golden, tmp_path):
    """drivers/test_ns_module (C host code over the GPU library) vs the reference driver semantics:
    int16 diff <= 1 LSB, almost all samples identical, extra stale frame reproduced."""
    from audiosignalprocess_amd.build import build_drivers

    exe = build_drivers()[0]
    pcm = golden["wav_in_i16"]
    hdr = b"RIFF" + struct.pack("<i", 36 + pcm.nbytes) + b"WAVE" + \
        struct.pack("<4sihhiihh", b"fmt ", 16, 1, 1, 16000, 32000, 2, 16) + \
        b"data" + struct.pack("<i", pcm.nbytes)
    src, dst = tmp_path / "in.wav", tmp_path / "out.wav"
    src.write_bytes(hdr + pcm.tobytes())
    subprocess.run([exe, str(src), str(dst), "-q"], check=True, stdout=subprocess.DEVNULL)
    raw = dst.read_bytes()
    assert raw[:44] == hdr  # header copied verbatim
    out = np.frombuffer(raw[44:], dtype=np.int16)
    ref = golden["wav_out_i16"]
    assert out.shape == ref.shape == (pcm.size + 160,)
    d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1
    assert (d == 0).mean() >= 0.999


def test_wav_driver_8khz(ns, golden8k, tmp_path):
    """drivers/test_ns_module on an 8 kHz mono WAV (test_ns_module.cpp:59-60: 80 samples per frame):
    the committed reference outputs of stream 0, rounded by FloatS16ToS16, within 1 LSB; the trailing
    frame of the `while (!feof)` loop (the stale last frame processed again) is present."""
    from audiosignalprocess_amd.build import build_drivers

    exe = build_drivers()[0]
    pcm = np.ascontiguousarray(golden8k["in_i16"][:, 0, :]).reshape(-1)
    hdr = b"RIFF" + struct.pack("<i", 36 + pcm.nbytes) + b"WAVE" + \
        struct.pack("<4sihhiihh", b"fmt ", 16, 1, 1, 8000, 16000, 2, 16) + \
        b"data" + struct.pack("<i", pcm.nbytes)
    src, dst = tmp_path / "in8k.wav", tmp_path / "out8k.wav"
    src.write_bytes(hdr + pcm.tobytes())
    subprocess.run([exe, str(src), str(dst), "-q"], check=True, stdout=subprocess.DEVNULL)
    raw = dst.read_bytes()
    assert raw[:44] == hdr
    out = np.frombuffer(raw[44:], dtype=np.int16)
    assert out.size == pcm.size + 80
    ref = golden8k["out_f32"][:, 0, :].reshape(-1)
    want = np.where(ref > 0, np.where(ref >= 32766.5, 32767, (ref + np.float32(0.5)).astype(np.int32)),
                    np.where(ref <= -32767.5, -32768, (ref - np.float32(0.5)).astype(np.int32)))
    d = np.abs(out[:pcm.size].astype(np.int32) - want)
    assert d.max() <= 1 and (d == 0).mean() >= 0.999


def _debug_compare(lib, fn_a, fn_b, start, count, param=1.0):
    n_bad = C.c_uint32()
    bad = (C.c_uint32 * 64)()
    lib.AspNs_debug_compare.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_float, C.c_int]
    assert lib.AspNs_debug_compare(fn_a, fn_b, start, count, C.byref(n_bad), bad, param, 0) == 0
    return n_bad.value, [bad[i] for i in range(min(n_bad.value, 64))]


@pytest.mark.parametrize("form", [0, 19])
def test_device_log_exhaustive_vs_libm_form(ns, form):
    """The kernels' lean (float)log((double)x) -- form 0: the series form (ns_device.h:
    log_f32_via_f64, also inside the AEC's pow), form 19: the table-driven form of the NS frame
    kernel (log_f32_via_tab) -- equals the libm-based form for EVERY positive float bit pattern
    (normal, denormal, inf): 2^31 inputs, checked on the device."""
    lib = ns.load_library()
    total_bad, examples = 0, []
    step = 1 << 28
    for start in range(0x00000000, 0x7f800001, step):
        count = min(step, 0x7f800001 - start)
        n, ex = _debug_compare(lib, form, 1, start, count)
        total_bad += n
        examples += ex
    assert total_bad == 0, [hex(b) for b in examples[:8]]


def test_device_libm_matches_host_libm(ns):
    """(float)log/exp/tanh((double)x) on the device == glibc on this host, over dense samples of
    the ranges the NS path feeds them (magn, 1+2 snr in [1, 2^31); -logLrt, lquantile in
    [-88, 88]; tanh arguments in [-40, 40])."""
    from tests.oracle_lib import oracle_lib

    lib, olib = ns.load_library(), oracle_lib()
    lib.AspNs_debug_eval.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    olib.asp_oracle_libm_f32.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(11)
    n = 1 << 22
    cases = {
        1: np.concatenate([rng.integers(0x3f800000, 0x4f000000, n, dtype=np.uint32).view(np.float32),
                           (1.0 + rng.random(n // 4) * 1e-3).astype(np.float32)]),
        2: ((rng.random(n) - 0.5) * 176).astype(np.float32),
        3: ((rng.random(n) - 0.5) * 80).astype(np.float32),
    }
    for fn, x in cases.items():
        dev = np.ascontiguousarray(x.copy())
        host = np.ascontiguousarray(x.copy())
        assert lib.AspNs_debug_eval(fn, dev.ctypes.data, dev.size, 0) == 0
        olib.asp_oracle_libm_f32(fn, host.ctypes.data, host.size)
        bad = np.nonzero(dev.view(np.uint32) != host.view(np.uint32))[0]
        assert bad.size == 0, (fn, bad.size, x[bad[:4]], dev[bad[:4]], host[bad[:4]])
    # and the lean log against glibc on the same sample
    dev = np.ascontiguousarray(cases[1].copy())
    host = np.ascontiguousarray(cases[1].copy())
    assert lib.AspNs_debug_eval(0, dev.ctypes.data, dev.size, 0) == 0
    olib.asp_oracle_libm_f32(1, host.ctypes.data, host.size)
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))


def test_device_division_forms_exact(ns):
    """The kernels' two cheaper division forms give the IEEE quotient: division by a wave-uniform
    divisor for every divisor the path uses (129, 50, counters 1..201, 500) and the lean Newton
    form, over full 2^23-mantissa sweeps of the other operand at several exponents."""
    lib = ns.load_library()

    def f32bits(v):
        return int(np.float32(v).view(np.uint32))

    sweeps = [f32bits(1.0), f32bits(2.0 ** 20), f32bits(2.0 ** -10)]
    divisors = [129.0, 50.0, 500.0, float(np.float32(0.1)), float(np.float32(0.05))] + [float(c) for c in range(1, 202)]
    for d in divisors:
        for start in sweeps[:2] if d not in (129.0, 50.0) else sweeps:
            n, ex = _debug_compare(lib, 4, 5, start, 1 << 23, d)
            assert n == 0, (d, [hex(b) for b in ex[:4]])
    rng = np.random.default_rng(2)
    params = list(np.exp(rng.uniform(np.log(1e-4), np.log(1e9), 40)).astype(np.float32)) + [1.0, 40.0, 1.0001]
    for prm in params:
        for start in sweeps:
            n, ex = _debug_compare(lib, 4, 6, start, 1 << 23, float(prm))   # x / param
            assert n == 0, ("x/p", prm, [hex(b) for b in ex[:4]])
            n, ex = _debug_compare(lib, 7, 8, start, 1 << 23, float(prm))   # param / x
            assert n == 0, ("p/x", prm, [hex(b) for b in ex[:4]])


def _sweep_all_floats(lib, fn_a, fn_b, lo=0x00000000, hi=0x100000000):
    total, examples = 0, []
    step = 1 << 28
    for start in range(lo, hi, step):
        n, ex = _debug_compare(lib, fn_a, fn_b, start, min(step, hi - start))
        total += n
        examples += ex
    return total, examples


def test_device_exp_tanh_sqrt_exhaustive(ns):
    """The kernels' lean exp / tanh equal the libm-based (float)f((double)x) forms, and the lean
    sqrt equals sqrtf, for EVERY float bit pattern (2^32 inputs each; sqrt: non-negative)."""
    lib = ns.load_library()
    for fa, fb, lo, hi, name in [(11, 2, 0, 1 << 32, "exp"), (12, 3, 0, 1 << 32, "tanh"),
                                 (10, 9, 0, 0x7f800001, "sqrt")]:
        bad, ex = _sweep_all_floats(lib, fa, fb, lo, hi)
        assert bad == 0, (name, bad, [hex(b) for b in ex[:8]])


def test_device_batched_forms_exhaustive(ns):
    """The N-argument forms the frame kernels use (straight-line fast paths, one merged fallback
    branch; ns_device.h: log_f32_via_tab_n / exp_f32_via_f64_n / fsqrt_n) against the libm forms for
    EVERY float in the first slot, with a second argument on the fast path and one that forces the
    fallback loop on every call (its result is checked inside the kernel)."""
    lib = ns.load_library()
    step = 1 << 28
    for fa, fb, lo, hi, params, name in [(20, 1, 0, 0x7f800001, (0.75, 1e-40), "log"),
                                         (21, 2, 0, 1 << 32, (-0.3, 100.0), "exp"),
                                         (22, 9, 0, 0x7f800001, (2.0, 1e-35), "sqrt")]:
        for prm in params:
            total, examples = 0, []
            for start in range(lo, hi, step):
                n, ex = _debug_compare(lib, fa, fb, start, min(step, hi - start), prm)
                total += n
                examples += ex
            assert total == 0, (name, prm, total, [hex(b) for b in examples[:8]])


def test_split_launch_is_identical(ns):
    """The fused step issued as 2..4 sub-launches on separate HIP streams gives the same bits."""
    S, F = 256, 30
    x = ns_frames(S, F, stream0=900)
    ref = ns.NsBatch(S, policy=1, kernel=1)
    y_ref = ref.analyze_process(x)
    for parts in (2, 3, 4):
        g = ns.NsBatch(S, policy=1, kernel=1)
        g.set_split(parts)
        assert np.array_equal(g.analyze_process(x), y_ref), parts
        assert state_diff(g.export_state(S - 1), ref.export_state(S - 1)) == {}
        g.close()
    ref.close()


def test_int16_pcm_path(ns, golden):
    """PCM in / PCM out fused into the kernel's loads and stores == float path + FloatS16ToS16,
    and within 1 LSB of the reference driver's WAV output."""
    pcm = golden["in_i16"]
    F, S, _ = pcm.shape
    g16 = ns.NsBatch(S, policy=1, kernel=1)
    y16 = g16.analyze_process_s16(pcm)
    gf = ns.NsBatch(S, policy=1, kernel=1)
    yf = gf.analyze_process(pcm.astype(np.float32))
    pos = np.where(yf >= np.float32(32766.5), 32767, (yf + np.float32(0.5)).astype(np.int32))
    neg = np.where(yf <= np.float32(-32767.5), -32768, (yf - np.float32(0.5)).astype(np.int32))
    assert np.array_equal(y16, np.where(yf > 0, pos, neg).astype(np.int16))
    ref = golden["wav_out_i16"][:F * 160].reshape(F, 160)
    d = np.abs(y16[:, 0].astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() >= 0.999
    assert state_diff(g16.export_state(1), gf.export_state(1)) == {}
    g16.close()
    gf.close()


def test_batched_wav_driver(ns, golden, tmp_path):
    """drivers/ns_batch_wav: several WAV files as one GPU batch; every output equals what the
    per-stream reference driver semantics give (incl. the stale extra frame), within 1 LSB."""
    from audiosignalprocess_amd.build import build_drivers

    exe = [e for e in build_drivers() if e.endswith("ns_batch_wav")][0]
    pcm = golden["wav_in_i16"]

    def wav(samples):
        return b"RIFF" + struct.pack("<i", 36 + samples.nbytes) + b"WAVE" + \
            struct.pack("<4sihhiihh", b"fmt ", 16, 1, 1, 16000, 32000, 2, 16) + \
            b"data" + struct.pack("<i", samples.nbytes) + samples.tobytes()

    (tmp_path / "in").mkdir()
    (tmp_path / "out").mkdir()
    lens = [pcm.size, 160 * 300 + 77, 160 * 41]
    for k, n in enumerate(lens):
        (tmp_path / "in" / ("s%d.wav" % k)).write_bytes(wav(pcm[:n]))
    subprocess.run([exe, str(tmp_path / "out")] + [str(tmp_path / "in" / ("s%d.wav" % k)) for k in range(3)],
                   check=True, stdout=subprocess.DEVNULL)
    ref_full = golden["wav_out_i16"]
    for k, n in enumerate(lens):
        out = np.frombuffer((tmp_path / "out" / ("s%d.wav" % k)).read_bytes()[44:], dtype=np.int16)
        frames = n // 160 + 1
        assert out.size == frames * 160
        # all but the last (stale / partial) frame are plain prefixes of the reference output
        m = (frames - 1) * 160
        d = np.abs(out[:m].astype(np.int32) - ref_full[:m].astype(np.int32))
        # start-up frames carry most of the reduction-order LSB flips (0.15 % of samples in the
        # first 41 frames, 0.02 % over 300), never more than 1 LSB
        assert d.max() <= 1 and (d == 0).mean() >= 0.997, k
    assert np.abs(out.astype(np.int32)).max() > 0


# ---------------------------------------------------------------------------------------------
# The one-stream-per-wave, two-bins-per-lane fused kernel: ns_kernels1.hip (kernel id 3, bin 128 on every
# lane: ASP_NS_REDUCE_TREE64P), and the two-streams-per-wave kernel ns_kernels2.hip (kernel id 2, four bins per
# lane of a half-wave: ASP_NS_REDUCE_TREE32): bit-exact against the oracle in the matching association,
# outputs and every state array.
PAIR_KERNELS = [(3, REDUCE_TREE64P), (2, REDUCE_TREE32)]

@pytest.mark.parametrize("kid,mode", PAIR_KERNELS)
def test_pair_kernel_free_running_bit_exact(ns, kid, mode):
    S, F = 24, 1100  # crosses blockInd 50 / 200 and two 500-frame histogram windows
    x = ns_frames(S, F, stream0=300)
    g = ns.NsBatch(S, policy=1, kernel=kid)
    y = g.analyze_process(x)
    o = OracleNs(S, policy=1, reduce_mode=mode)
    yo = o.run(x, threads=8)
    assert np.isfinite(y).all()
    bad = np.nonzero((y != yo).any(axis=2))
    assert bad[0].size == 0, (bad[0][:5], bad[1][:5])
    for s in range(0, S, 5):
        assert state_diff(g.export_state(s), o.export_state(s)) == {}
    g.close()


@pytest.mark.parametrize("kid,mode", PAIR_KERNELS)
def test_pair_kernel_edge_cases_policies_and_odd_count(ns, kid, mode):
    S, F = 7, 260
    x = ns_frames(S, F, stream0=40)
    x[10:16, 1] = 0.0      # energy == 0 frames in the middle of a run
    x[:, 2] = 0.0          # digital silence from the start
    x[30:, 3] = 32767.0    # full-scale DC
    x[100:104, 4] = 0.0
    for policy in (0, 1, 2, 3):
        g = ns.NsBatch(S, policy=policy, kernel=kid)
        y = g.analyze_process(x)
        o = OracleNs(S, policy=policy, reduce_mode=mode)
        yo = o.run(x)
        assert np.array_equal(y, yo), policy
        for s in range(S):
            assert state_diff(g.export_state(s), o.export_state(s)) == {}, (policy, s)
        g.close()


@pytest.mark.parametrize("kid,mode", PAIR_KERNELS)
def test_pair_kernel_golden_int16_split_and_scale(ns, golden, kid, mode):
    pcm = golden["in_i16"]
    F, S, _ = pcm.shape
    g = ns.NsBatch(S, policy=1, kernel=kid)
    y = g.analyze_process(pcm.astype(np.float32))
    rel = rel_l2_per_stream(y, golden["out_f32"])
    check_free_running(rel)
    g16 = ns.NsBatch(S, policy=1, kernel=kid)
    y16 = g16.analyze_process_s16(pcm)
    ref = golden["wav_out_i16"][:F * 160].reshape(F, 160)
    d = np.abs(y16[:, 0].astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() >= 0.999
    g.close()
    g16.close()
    # BASELINE config[1] size, 4 sub-launches, and config 5's per-GPU size, spot-checked against the oracle
    for S2, parts in ((4096, 4), (8192, 2)):
        F2 = 40
        base = ns_frames(16, F2, stream0=11)
        idx = np.arange(S2) % 16
        x = np.ascontiguousarray(base[:, idx])
        big = ns.NsBatch(S2, policy=1, kernel=kid)
        big.set_split(parts)
        yb = big.analyze_process(x)
        assert np.isfinite(yb).all()
        yo = OracleNs(16, policy=1, reduce_mode=mode).run(base)
        for k in (0, 1, 2, 3, 1364, 1365, 2730, 2731, S2 - 2, S2 - 1):
            assert np.array_equal(yb[:, k], yo[:, idx[k]]), (S2, k)
        big.close()


@pytest.mark.parametrize("kid,mode", PAIR_KERNELS)
def test_pair_then_unfused_continues(ns, kid, mode):
    """Fused one-per-wave (pair layout) steps, then the reference's two-call protocol on the same batch."""
    S, F = 5, 130
    x = ns_frames(S, F, stream0=77)
    g = ns.NsBatch(S, policy=1, kernel=kid)
    o = OracleNs(S, policy=1, reduce_mode=mode)
    assert np.array_equal(g.analyze_process(x[:100]), o.run(x[:100]))
    o.mode = REDUCE_TREE  # separate Analyze / Process launches use the q / q + 64 kernels
    for f in range(100, F):
        g.analyze(x[f])
        o.analyze(x[f])
        assert np.array_equal(g.process(x[f]), o.process(x[f])), f
    g.close()


# ---------------------------------------------------------------------------------------------
# 8 kHz (ns_core.c:89-98: blockLen 80, anaLen 128, 65 bins, kBlocks80w128, WebRtc_rdft(128)): the G8
# instantiation of ns_kernels.hip behind every entry point of a batch initialised at fs = 8000.
@pytest.fixture(scope="module")
def golden8k():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "ns8k_golden.npz")))


def test_8khz_rdft128_bit_exact(ns, golden8k):
    from audiosignalprocess_amd.ns import rdft128

    # value equality, as for the 256-point seam: an exact zero may carry the other sign where a
    # twiddle-free butterfly multiplies by (1, 0) instead of skipping the product (impulse inputs only)
    fwd = rdft128(golden8k["fft_in"], 1)
    assert np.array_equal(fwd, golden8k["fft_fwd"])
    assert np.array_equal(rdft128(golden8k["fft_fwd"], -1), golden8k["fft_inv"])
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((300, 128)) * 5000).astype(np.float32)
    o = OracleNs(1, fs=8000)
    for isgn in (1, -1):
        assert np.array_equal(rdft128(x, isgn).view(np.uint32), o.rdft128(x, isgn).view(np.uint32)), isgn


def test_8khz_free_running_bit_exact_all_policies(ns):
    S, F = 13, 560   # crosses blockInd 50 / 200 and a 500-frame histogram window
    x = ns_frames(S, F, stream0=90)[:, :, ::2].copy()
    x[20:24, 1] = 0.0
    x[:, 2] = 0.0
    x[60:, 3] = 32767.0
    for policy in (1, 0, 3):
        g = ns.NsBatch(S, fs=8000, policy=policy)
        o = OracleNs(S, policy=policy, reduce_mode=REDUCE_TREE, fs=8000)
        y, yo = g.analyze_process(x), o.run(x, threads=4)
        bad = np.nonzero((y != yo).any(axis=2))
        assert bad[0].size == 0, (policy, bad[0][:5], bad[1][:5])
        for s in range(S):
            assert state_diff(g.export_state(s), o.export_state(s)) == {}, (policy, s)
        g.close()


def test_8khz_golden_protocols_and_pcm(ns, golden8k):
    """The reference's own 8 kHz outputs within the 1e-4 bar; Analyze + Process as two calls equals the
    fused step; the int16 entry point within 1 LSB of FloatS16ToS16 of the reference's floats; state
    import / export round trip; the reference's layer-1 symbols at fs = 8000."""
    x = golden8k["in_i16"].astype(np.float32)
    F, S, _ = x.shape
    g = ns.NsBatch(S, fs=8000, policy=1)
    y = g.analyze_process(x)
    check_free_running(rel_l2_per_stream(y, golden8k["out_f32"]), "8 kHz HIP vs reference golden")
    # two-call protocol on a second batch, switching over mid-run
    u = ns.NsBatch(S, fs=8000, policy=1)
    assert np.array_equal(u.analyze_process(x[:120]), y[:120])
    for f in range(120, 260):
        u.analyze(x[f])
        assert np.array_equal(u.process(x[f]), y[f]), f
    # checkpoint: export from one batch, import into a fresh one, continue
    v = ns.NsBatch(S, fs=8000, policy=1)
    for s in range(S):
        v.import_state(s, u.export_state(s))
    assert np.array_equal(v.analyze_process(x[260:300]), y[260:300])
    # PCM in / out
    g16 = ns.NsBatch(S, fs=8000, policy=1)
    y16 = g16.analyze_process_s16(golden8k["in_i16"])
    ref = golden8k["out_f32"]
    want = np.where(ref > 0, np.where(ref >= 32766.5, 32767, (ref + np.float32(0.5)).astype(np.int32)),
                    np.where(ref <= -32767.5, -32768, (ref - np.float32(0.5)).astype(np.int32)))
    d = np.abs(y16.astype(np.int32) - want)
    assert d.max() <= 1 and (d == 0).mean() >= 0.999
    for b in (g, u, v, g16):
        b.close()
    # layer 1: WebRtcNs_Create / _Init(8000) / _set_policy / _Analyze / _Process (noise_suppression.h:35-123)
    lib = ns.load_library()
    h = C.c_void_p()
    assert lib.WebRtcNs_Create(C.byref(h)) == 0
    lib.WebRtcNs_Init.argtypes = [C.c_void_p, C.c_uint32]
    assert lib.WebRtcNs_Init(h, 8000) == 0 and lib.WebRtcNs_set_policy(h, 1) == 0
    lib.WebRtcNs_Analyze.argtypes = [C.c_void_p, C.c_void_p]
    lib.WebRtcNs_Process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.WebRtcNs_Analyze.restype = lib.WebRtcNs_Process.restype = None
    o = OracleNs(1, policy=1, reduce_mode=REDUCE_TREE, fs=8000)
    for f in range(60):
        fr = np.ascontiguousarray(x[f, 0])
        out = np.empty(80, np.float32)
        pin, pout = (C.c_void_p * 1)(fr.ctypes.data), (C.c_void_p * 1)(out.ctypes.data)
        lib.WebRtcNs_Analyze(h, fr.ctypes.data)
        lib.WebRtcNs_Process(h, pin, 1, pout)
        o.analyze(fr[None])
        assert np.array_equal(out, o.process(fr[None])[0]), f
    lib.WebRtcNs_Free.argtypes = [C.c_void_p]
    lib.WebRtcNs_Free(h)


# ---------------------------------------------------------------------------------------------
# The timed entry point of bench.py (AspNsBatch_TimedSteps: K steps over a device ring, launch chains)
# against the oracle: same launches as AnalyzeProcess, but the ring wrap-around (step k uses slot
# k % ring) and the split into chains only run here.
@pytest.mark.parametrize("S,split,flow,kid", [(16, 2, 0, 3), (4096, 2, 0, 3), (4100, 3, 0, 3), (16, 1, 1, 3), (4096, 1, 1, 3),
                                              (4100, 1, 1, 3), (9001, 1, 1, 3), (16, 1, 1, 2), (4096, 1, 1, 2), (9001, 1, 1, 2)])
def test_timed_steps_ring_wraparound_bit_exact_vs_oracle(ns, S, split, flow, kid):
    from audiosignalprocess_amd.ns import DeviceBuffer

    ring, steps = 7, 60          # ring < steps: every slot is reused eight times
    D = 16                       # distinct streams behind the batch
    base = ns_frames(D, ring, stream0=21)
    idx = np.arange(S) % D
    x = np.ascontiguousarray(base[:, idx])
    din, dout = DeviceBuffer(x.nbytes), DeviceBuffer(x.nbytes)
    din.upload(x)
    g = ns.NsBatch(S, policy=1, kernel=kid)   # fresh batch: the run crosses blockInd 50
    g.set_split(split)
    g.set_flow(flow)             # 0: launch chains; 1: the hand-off build (one launch per step, steps overlap)
    ms = g.timed_steps(din.ptr, dout.ptr, ring, steps)
    assert ms > 0
    got = dout.download(x.shape)
    o = OracleNs(D, policy=1, reduce_mode=REDUCE_TREE64P if kid == 3 else REDUCE_TREE32)
    want = np.empty((ring, D, 160), np.float32)
    for k in range(steps):
        want[k % ring] = o.run(base[k % ring:k % ring + 1])[0]   # slot k % ring keeps the last step that used it
    spots = range(S) if S <= 64 else (0, 1, 5, 1364, 1365, 2047, 2048, 2049, 2730, S - 2, S - 1)
    for k in spots:
        assert np.array_equal(got[:, k], want[:, idx[k]]), k
    for k in (0, S // 2 - 1, S // 2, S - 1):
        assert state_diff(g.export_state(k), o.export_state(int(idx[k]))) == {}, k
    g.close()


# ---------------------------------------------------------------------------------------------
# The hand-off build (AspNsBatch_SetFlow; the default of every multi-frame entry point) against the plain
# launches of the same kernel: outputs and full state bit for bit over 520 steps -- through both start-up
# windows and the first close of the histogram window (frame 500) -- with a silent stream (the early exit
# publishes too), stream counts that leave the last workgroup ragged, and batches of more streams than the
# chip holds waves (a wave then walks 2-4 streams per step).  Two hand-off batches run at the same time
# (their launches interleave on the chip), so every wait happens under uneven load.
@pytest.mark.parametrize("S,kid", [(5, 3), (4100, 3), (8192, 3), (12290, 3), (5, 2), (4101, 2), (8192, 2), (12290, 2)])
def test_handoff_build_equals_plain_launches(ns, S, kid):
    from audiosignalprocess_amd.ns import DeviceBuffer

    ring, steps, D = 9, 520, 16
    base = ns_frames(D, ring, stream0=40)
    base[:, 3] = 0.0                      # a silent stream: zero-energy exit every frame
    base[4:, 7] = 0.0                     # and one that falls silent part of the time
    idx = (np.arange(S) * 7) % D
    x = np.ascontiguousarray(base[:, idx])
    din = DeviceBuffer(x.nbytes)
    din.upload(x)
    outs, batches = [], []
    for flow in (0, 1, 1):
        g = ns.NsBatch(S, policy=2, kernel=kid)   # 3: one stream per wave (pair layout); 2: two streams per wave
        g.set_flow(flow)
        g.set_split(2 if flow == 0 else 1)
        dout = DeviceBuffer(x.nbytes)
        g.analyze_process_replay(din.ptr, dout.ptr, ring, steps)   # asynchronous: the batches overlap
        outs.append(dout)
        batches.append(g)
    for g in batches:
        g.synchronize()
    y = [o.download(x.shape) for o in outs]
    assert np.isfinite(y[0]).all() and np.abs(y[0]).max() > 0
    for k in (1, 2):
        assert np.array_equal(y[0].view(np.uint32), y[k].view(np.uint32)), k
    spots = sorted(set([0, 1, 3, 4, S // 2, S - 2, S - 1] + list(range(0, S, max(1, S // 37)))))
    for s_ in spots:
        want = batches[0].export_state(s_)
        for k in (1, 2):
            assert state_diff(batches[k].export_state(s_), want) == {}, (s_, k)
    for g in batches:
        g.close()


def test_per_stream_init_and_policy_vs_one_oracle_per_stream(ns):
    """The reference takes Init and set_policy per handle (noise_suppression.c:35-44).  A batch whose streams run
    four different policies, one of them re-initialised in the middle of the run and given another policy a few
    frames later: every stream bit-equal (outputs and state) to its OWN oracle driven the same way."""
    S, F = 8, 260
    x = ns_frames(S, F, stream0=9)
    g = ns.NsBatch(S)                      # policy 0 after Init
    pol = [0, 1, 2, 3, 1, 2, 3, 0]
    for s_, m in enumerate(pol):
        g.set_policy_stream(s_, m)
    cuts = [0, 70, 77, 180, F]            # re-init stream 5 at frame 70, policy 3 at 77, re-init stream 0 at 180
    y = np.empty_like(x)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == 70:
            g.init_stream(5)
        if a == 77:
            g.set_policy_stream(5, 3)
        if a == 180:
            g.init_stream(0)
        y[a:b] = g.analyze_process(x[a:b])
    for s_ in range(S):
        o = OracleNs(1, policy=pol[s_], reduce_mode=REDUCE_TREE64P)
        xs = np.ascontiguousarray(x[:, s_:s_ + 1])
        if s_ == 5:
            want = o.run(xs[:70])
            o = OracleNs(1, policy=0, reduce_mode=REDUCE_TREE64P)
            w2 = o.run(xs[70:77])
            o.set_policy(3)
            w3 = o.run(xs[77:])
            want = np.concatenate([want, w2, w3], axis=0)
        elif s_ == 0:
            w1 = o.run(xs[:180])
            o = OracleNs(1, policy=0, reduce_mode=REDUCE_TREE64P)
            want = np.concatenate([w1, o.run(xs[180:])], axis=0)
        else:
            want = o.run(xs)
        assert np.array_equal(y[:, s_], want[:, 0]), s_
        assert state_diff(g.export_state(s_), o.export_state(0)) == {}, s_
    g.close()


def test_handoff_wait_times_out_loudly(ns):
    """A hand-off launch whose predecessor never ran must not hang: the bounded wait gives up, the grid
    drains, the call reports ASP_ERR_HIP, and the batch works again after Init."""
    S = 64
    x = ns_frames(S, 6, stream0=3)
    g = ns.NsBatch(S, policy=1)
    g.set_flow(1)
    want = g.analyze_process(x)
    g.lib.AspNsBatch_DebugFlowDesync(g.h)
    with pytest.raises(RuntimeError):
        g.analyze_process(x)
    g.init(16000)
    g.set_policy(1)
    assert np.array_equal(g.analyze_process(x), want)
    g.close()


# ---------------------------------------------------------------------------------------------
# SURVEY 8(c)(3) on the device: the default kernel, free running, against OracleNs(SEQ) -- which equals
# the compiled reference bit for bit (tests/test_ns_oracle.py) -- over 64 streams x 1100 frames:
# median and 95th percentile of the per-stream relative L2 within 1e-4, the maximum with its stream and
# the first frame beyond 1e-4 printed.  The maximum is held to 1e-4 as well; CHAOS_CAP (the reference's
# own self-distance under other compiler flags) is the documented ceiling it may never pass.
@pytest.mark.parametrize("S", [64, 8192])
def test_default_kernel_vs_reference_percentiles(ns, S):
    D, F = 64, 1100
    x64 = ns_frames(D, F, stream0=0)
    ref = OracleNs(D, policy=1, reduce_mode=REDUCE_SEQ).run(x64, threads=8)
    if S == D:
        y = ns.NsBatch(S, policy=1).analyze_process(x64)
    else:
        # the 8192-stream share of BASELINE config 5: the 64 streams replicated, 64 spot streams read back
        idx = np.arange(S) % D
        g = ns.NsBatch(S, policy=1)
        g.set_split(2)
        yb = np.empty((F, D, 160), np.float32)
        pick = (np.arange(D) * 127 + 5) % S
        for f0 in range(0, F, 100):
            blk = g.analyze_process(np.ascontiguousarray(x64[f0:f0 + 100][:, idx]))
            yb[f0:f0 + 100] = blk[:, pick]
        ref = ref[:, idx[pick]]
        y = yb
        g.close()
    rel, worst, first = free_running_report(y, ref, "HIP default kernel, %d streams, vs the reference-equal oracle" % S)
    assert np.median(rel) <= 1e-6
    assert np.percentile(rel, 95) <= 1e-4
    assert rel.max() <= CHAOS_CAP            # the documented ceiling ...
    assert rel.max() <= 1e-4, (worst, first)  # ... and today's measurement stays inside the 1e-4 bar


# ---------------------------------------------------------------------------------------------
# libapm's APM_NS class (include/apm_ns.h over the C-ABI): channels of an interleaved capture
# stream are the streams of one batch.
def test_apm_ns_class_interleaved_capture(ns, golden, tmp_path):
    """drivers/apm_ns_raw drives APM_NS::processCaptureStream (short and float overloads) on a
    3-channel interleaved capture; equals the per-channel int16 path bit for bit, with the
    FloatToS16 / S16ToFloat conversions of audio_util.h:27-39 around it for float input."""
    from audiosignalprocess_amd.build import build_drivers

    exe = [e for e in build_drivers() if e.endswith("apm_ns_raw")][0]
    pcm = golden["wav_in_i16"]
    F, C3 = 120, 3
    planar = np.stack([pcm[o:o + F * 160] for o in (0, 4000, 9000)])           # [C][F*160]
    inter = np.ascontiguousarray(planar.T)                                     # [F*160][C]
    for mode in (0, 2):
        b = ns.NsBatch(C3, policy=mode)
        want = b.analyze_process_s16(planar.reshape(C3, F, 160).transpose(1, 0, 2))  # [F][C][160]
        want_inter = want.transpose(0, 2, 1).reshape(F * 160, C3)
        (tmp_path / "in.s16").write_bytes(inter.tobytes())
        subprocess.run([exe, str(tmp_path / "in.s16"), str(tmp_path / "out.s16"), str(C3), str(mode), "s16"],
                       check=True)
        got = np.frombuffer((tmp_path / "out.s16").read_bytes(), np.int16).reshape(F * 160, C3)
        assert np.array_equal(got, want_inter), mode
        assert not np.array_equal(got, inter)

    # float overload: x in [-1, 1]; choose x = s / 32768 so FloatToS16 is easy to restate
    x = (inter.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    v = x
    pos = np.where(v >= 1, 32767, (v * np.float32(32767) + np.float32(0.5)).astype(np.float32)).astype(np.int64)
    neg = np.where(v <= -1, -32768, (-v * np.float32(-32768) - np.float32(0.5)).astype(np.float32)).astype(np.int64)
    s16 = np.where(v > 0, pos, neg).astype(np.int16)                           # trunc toward zero
    b = ns.NsBatch(C3, policy=1)
    den = b.analyze_process_s16(np.ascontiguousarray(s16.T).reshape(C3, F, 160).transpose(1, 0, 2))
    den = den.transpose(0, 2, 1).reshape(F * 160, C3)
    k_max = np.float32(1.0) / np.float32(32767)
    k_min = np.float32(1.0) / np.float32(-32768)
    want_f = den.astype(np.float32) * np.where(den > 0, k_max, -k_min).astype(np.float32)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    subprocess.run([exe, str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), str(C3), "1", "f32"], check=True)
    got_f = np.frombuffer((tmp_path / "out.f32").read_bytes(), np.float32).reshape(F * 160, C3)
    assert np.array_equal(got_f, want_f)


# ---------------------------------------------------------------------------------------------
# 32 / 48 kHz: high-band branch (ns_kernels_hb.hip) next to the unchanged low-band kernels
def _hb_frames(S, F, nh):
    from tests.test_ns_oracle import _band_frames
    return _band_frames(S, F, nh)


@pytest.mark.parametrize("fs,nh,spw", [(32000, 1, 3), (48000, 2, 3), (32000, 1, 1)])
def test_high_band_bit_exact_vs_oracle(ns, fs, nh, spw):
    """Low band as before (bit-exact vs the TREE oracle of the kernel in use); the high-band gain is
    computed from that state with the reference's own summation order, so the high-band outputs and
    the carried buffers are bit-exact too, through start-up, model updates, zero-energy frames and
    saturation."""
    S, F = 6, 560
    low, high = _hb_frames(S, F, nh)
    g = ns.NsBatch(S, fs=fs, policy=2, kernel=spw)
    o = OracleNs(S, policy=2, reduce_mode=REDUCE_TREE64P if spw == 3 else REDUCE_TREE, fs=fs)
    gl, gh = g.analyze_process_bands(low, high)
    ol, oh = o.run_bands(low, high)
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32))
    assert np.array_equal(gh.view(np.uint32), oh.view(np.uint32))
    for s in range(S):
        hb = np.ctypeslib.as_array(o.hb[s].dataBufHB).reshape(2, 256)
        assert np.array_equal(g.export_hb(s)[:nh, 160:], hb[:nh, 160:])
    assert np.abs(gh[450:]).mean() < 0.9 * np.abs(high[450:]).mean()


def test_high_band_golden_and_unfused_protocol(ns):
    """The reference's own 32 kHz outputs (committed fixture): low band within the 1e-4 bar, high
    band within 1e-4 as well (its gain depends on the low band's state); and the two-call protocol
    WebRtcNs_Analyze + WebRtcNs_Process(bands) gives exactly what the fused entry point gives."""
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "ns_hb_golden.npz")))
    low, high = gold["low_i16"].astype(np.float32), gold["high_i16"].astype(np.float32)[:, None]
    S = low.shape[1]
    g = ns.NsBatch(S, fs=32000, policy=2)
    gl, gh = g.analyze_process_bands(low, high)
    assert rel_l2_per_stream(gl, gold["out_low"]).max() <= 1e-4
    assert rel_l2_per_stream(gh[:, 0], gold["out_high"]).max() <= 1e-4
    u = ns.NsBatch(S, fs=32000, policy=2, kernel=1)
    f = ns.NsBatch(S, fs=32000, policy=2, kernel=1)
    fl, fh = f.analyze_process_bands(low[:130], high[:130])
    for k in range(130):
        u.analyze(low[k])
        ul, uh = u.process_bands(low[k], high[k])
        assert np.array_equal(ul.view(np.uint32), fl[k].view(np.uint32)), k
        assert np.array_equal(uh.view(np.uint32), fh[k].view(np.uint32)), k


def test_one_band_entry_points_refuse_multi_band_batches(ns):
    """A batch initialised at 32 / 48 kHz carries a high-band delay line (ns_core.c:1227-1235) that only
    the bands entry points advance: the one-band ones fail with ASP_ERR_STATE instead of silently
    letting it fall out of step."""
    S = 4
    x = ns_frames(S, 2)
    for fs in (32000, 48000):
        g = ns.NsBatch(S, fs=fs, policy=1)
        with pytest.raises(ns.AspError):
            g.analyze_process(x)
        with pytest.raises(ns.AspError):
            g.analyze_process_s16(x.astype(np.int16))
        g.close()


@pytest.mark.parametrize("freq", [32000, 48000])
def test_apm_ns_class_band_split_rates(ns, golden, tmp_path, freq):
    """APM_NS at 32 / 48 kHz (include/apm_ns.h): interleaved capture -> SplittingFilter analysis (QMF;
    at 48 kHz the sinc resampler and two more QMF stages) -> suppressor with one / two high bands ->
    synthesis, per 10 ms; equals the same chain assembled from the batch entry points (each of which
    is checked against its oracle) bit for bit."""
    from audiosignalprocess_amd.build import build_drivers
    from audiosignalprocess_amd.qmf import SplitBatch

    exe = [e for e in build_drivers() if e.endswith("apm_ns_raw")][0]
    pcm = golden["wav_in_i16"]
    nb = freq // 16000
    n = 160 * nb
    F, C2 = 90, 2
    # a stereo capture at `freq`: two offsets of the fixture, up-sampled by sample repetition plus a
    # little dither so the high bands are not empty
    rng = np.random.default_rng(2)
    planar = np.stack([np.repeat(pcm[o:o + F * 160], nb) for o in (0, 7000)]).astype(np.int32)
    planar = np.clip(planar + rng.integers(-200, 200, planar.shape), -32768, 32767).astype(np.int16)
    inter = np.ascontiguousarray(planar.T)                                      # [F*n][C]
    (tmp_path / "in.s16").write_bytes(inter.tobytes())
    subprocess.run([exe, str(tmp_path / "in.s16"), str(tmp_path / "out.s16"), str(C2), "1", "s16", str(freq)],
                   check=True)
    got = np.frombuffer((tmp_path / "out.s16").read_bytes(), np.int16).reshape(F * n, C2)

    def s16(v):   # FloatS16ToS16, audio_util.h:41-49
        r = np.where(v > 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5)))
        return np.clip(r, -32768, 32767).astype(np.int16)

    q, b = SplitBatch(C2, nb), ns.NsBatch(C2, fs=freq, policy=1)
    want = np.empty((F, C2, n), np.int16)
    frames = planar.reshape(C2, F, n).transpose(1, 0, 2)
    for f in range(F):
        bands = q.analysis(frames[f]).astype(np.float32)                       # [nb][C][160]
        ol, oh = b.analyze_process_bands(bands[0][None], bands[1:][None])
        want[f] = q.synthesis(np.concatenate([s16(ol), s16(oh[0])]))
    want_inter = want.transpose(0, 2, 1).reshape(F * n, C2)
    assert np.array_equal(got, want_inter)
    assert not np.array_equal(got, inter)


def test_apm_ns_48khz_float_vs_reference_libapm(ns, tmp_path):
    """The scenario of test_libapm/test_apm_ns_float.cpp against the reference's own libapm class
    (committed fixture, built from APM_NS + AudioBuffer + SplittingFilter + sinc resampler + float
    suppressor compiled in place): 48 kHz stereo float capture through include/apm_ns.h on the GPU.
    Everything around the suppressor is integer-exact; the suppressor's ~10 cross-bin sums per
    frame are associated differently, so a small share of output samples sits
    one int16 step away."""
    from audiosignalprocess_amd.build import build_drivers

    exe = [e for e in build_drivers() if e.endswith("apm_ns_raw")][0]
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "apm_golden.npz")))
    x = (gold["in_i16"].astype(np.float32) / np.float32(32768.0)).astype(np.float32)   # [F][480][2]
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    subprocess.run([exe, str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), "2", "1", "f32", "48000"], check=True)
    got = np.frombuffer((tmp_path / "out.f32").read_bytes(), np.float32).reshape(gold["out_f32"].shape)
    want = gold["out_f32"]
    # outputs are S16ToFloat(int16): compare on the int16 grid
    gi = np.rint(got.astype(np.float64) * np.where(got > 0, 32767.0, 32768.0)).astype(np.int32)
    wi = np.rint(want.astype(np.float64) * np.where(want > 0, 32767.0, 32768.0)).astype(np.int32)
    d = np.abs(gi - wi)
    parity_note("APM_NS 48 kHz stereo float: %.4f of samples identical, max |diff| %d LSB" % ((d == 0).mean(), d.max()))
    assert d.max() <= 2 and (d == 0).mean() >= 0.99
    assert rel_l2_per_stream(got.reshape(-1, 1, 960), want.reshape(-1, 1, 960)).max() <= 1e-4
