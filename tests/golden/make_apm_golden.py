#!/usr/bin/env python3
"""Generates tests/golden/apm_golden.npz from the REFERENCE libapm build (oracle/_ref/libapm_ref.so:
APM_NS over AudioBuffer, SplittingFilter, the sinc resampler and the float suppressor, all compiled
in place).  Build container only.  The scenario of test_libapm/test_apm_ns_float.cpp: 48 kHz stereo
float capture, 10 ms at a time, mode Ns_Mode_Mideum.
  in_i16 [F][480][2] int16: the capture is in_i16 / 32768 (float32);  out_f32 [F][480][2] float32."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import RefApm, have_apm_ref  # noqa: E402

F = 80


def capture():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ns_golden.npz")))
    pcm = g["wav_in_i16"]
    rng = np.random.default_rng(4)
    planar = np.stack([np.repeat(pcm[o:o + F * 160], 3) for o in (0, 7000)]).astype(np.int32)
    planar = np.clip(planar + rng.integers(-150, 150, planar.shape), -32768, 32767).astype(np.int16)
    return np.ascontiguousarray(planar.T).reshape(F, 480, 2)


def main():
    assert have_apm_ref(), "build oracle/_ref first (make -C oracle)"
    x = capture()
    ref = RefApm(48000, 1, 2)
    out = np.stack([ref.process_f32(fr.astype(np.float32) / np.float32(32768.0)) for fr in x])
    path = os.path.join(ROOT, "tests", "golden", "apm_golden.npz")
    np.savez_compressed(path, in_i16=x, out_f32=out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
