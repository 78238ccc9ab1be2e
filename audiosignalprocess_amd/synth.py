"""Deterministic synthetic inputs; numpy only.

These are the generators the bench and the parity tests feed to every engine
(HIP path, CPU restatement, compiled reference) so all of them see identical
float32 frames.
"""
import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


def _lcg_uniform(seeds, t0, count):
    """n[s, k] for absolute sample indices t0 .. t0+count-1 of each stream.

    u <- u*1664525 + 1013904223 (mod 2**32) once per sample, seeded per
    stream; n = ((u >> 16) - 32768) / 32768 in [-1, 1).
    """
    seeds = np.asarray(seeds, dtype=np.uint32)
    total = t0 + count
    with np.errstate(over="ignore"):
        a_pow = np.cumprod(np.full(total, _A, dtype=np.uint32), dtype=np.uint32)  # a^(t+1)
        geo = np.cumsum(np.concatenate(([np.uint32(1)], a_pow[:-1])), dtype=np.uint32)
        c_t = (_C * geo).astype(np.uint32)  # c * (a^t + ... + 1)
        u = seeds[:, None] * a_pow[None, t0:] + c_t[None, t0:]
    u = u.astype(np.uint32)
    return ((u >> np.uint32(16)).astype(np.float64) - 32768.0) / 32768.0


def ns_frames(num_streams, num_frames, stream0=0, frame0=0):
    """NS input, float-S16 units, shape [num_frames][num_streams][160] float32.

    x = 600 n + g(f) 3000 sin(2 pi (300 + 5 (s mod 64)) t / 16000) (0.5 + 0.5 sin(0.01 f)),
    g(f) = 1 when floor(f/100) is odd else 0; never all-zero.
    """
    s = np.arange(stream0, stream0 + num_streams, dtype=np.int64)
    seeds = (12345 + 7919 * s) & 0xFFFFFFFF
    n = _lcg_uniform(seeds, 160 * frame0, 160 * num_frames)  # [S][T]
    t = np.arange(160 * frame0, 160 * (frame0 + num_frames), dtype=np.float64)
    f = np.floor(t / 160.0)
    g = ((np.floor(f / 100.0).astype(np.int64) & 1) == 1).astype(np.float64)
    freq = 300.0 + 5.0 * (s % 64).astype(np.float64)
    tone = np.sin(2.0 * np.pi * freq[:, None] * t[None, :] / 16000.0)
    env = g * 3000.0 * (0.5 + 0.5 * np.sin(0.01 * f))
    x = 600.0 * n + tone * env[None, :]
    x = x.astype(np.float32).reshape(num_streams, num_frames, 160)
    return np.ascontiguousarray(x.transpose(1, 0, 2))


def bt_samples(num_streams, num_samples, stream0=0, t0=0):
    """BlockThresholding input, float in [-1, 1], shape [num_streams][num_samples] float32.

    0.2 sin(0.02 t (1 + 0.1 (s mod 16))) (1 if floor(t/20000) odd else 0.1) + 0.08 n
    (SURVEY.md section 8(d)); never an all-zero macroblock (the reference divides by the
    block energy, audioDenoiseBlockTreshold.c:397-398).
    """
    s = np.arange(stream0, stream0 + num_streams, dtype=np.int64)
    seeds = (12345 + 7919 * s) & 0xFFFFFFFF
    n = _lcg_uniform(seeds, t0, num_samples)
    t = np.arange(t0, t0 + num_samples, dtype=np.float64)
    gate = np.where((np.floor(t / 20000.0).astype(np.int64) & 1) == 1, 1.0, 0.1)
    w = 0.02 * (1.0 + 0.1 * (s % 16).astype(np.float64))
    x = 0.2 * np.sin(w[:, None] * t[None, :]) * gate[None, :] + 0.08 * n
    return np.ascontiguousarray(x.astype(np.float32))


def aec_frames(num_streams, num_frames, stream0=0):
    """AEC input (far, near), float-S16 units, each [num_frames][num_streams][160] float32.

    far = A(f) n_far, A = 4000 when floor(f/150) is odd else 40;
    near = 0.5 far[t-40-(s mod 16)] + 0.25 far[t-90] + 100 n_near
           (+ 1500 sin(0.05 t) when floor(f/200) mod 3 == 2)        (SURVEY.md section 8(d)).
    """
    s = np.arange(stream0, stream0 + num_streams, dtype=np.int64)
    T = 160 * num_frames
    n_far = _lcg_uniform((12345 + 7919 * s) & 0xFFFFFFFF, 0, T)
    n_near = _lcg_uniform((987654321 + 104729 * s) & 0xFFFFFFFF, 0, T)
    t = np.arange(T, dtype=np.float64)
    f = np.floor(t / 160.0).astype(np.int64)
    amp = np.where((f // 150) % 2 == 1, 4000.0, 40.0)
    far = n_far * amp[None, :]
    near = 100.0 * n_near
    for k in range(num_streams):
        d1 = 40 + int(s[k] % 16)
        near[k, d1:] += 0.5 * far[k, :-d1]
        near[k, 90:] += 0.25 * far[k, :-90]
    talk = ((f // 200) % 3 == 2)
    near += (1500.0 * np.sin(0.05 * t) * talk)[None, :]

    def shape(a):
        return np.ascontiguousarray(a.astype(np.float32).reshape(num_streams, num_frames, 160).transpose(1, 0, 2))

    return shape(far), shape(near)
