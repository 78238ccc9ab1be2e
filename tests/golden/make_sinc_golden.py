#!/usr/bin/env python3
"""Generates tests/golden/sinc_golden.npz from the REFERENCE sinc resampler build
(oracle/_ref/libsinc_ref.so).  Build container only.
  x48 [F][480] int16 -> y64 [F][640] (PushSincResampler(480, 640)) -> z48 [F][480] (640 -> 480)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import RefSinc, have_sinc_ref  # noqa: E402
from tests.test_sinc_oracle import sinc_inputs  # noqa: E402


def main():
    assert have_sinc_ref(), "build oracle/_ref first (make -C oracle)"
    x = sinc_inputs(30, 480)
    up, down = RefSinc(480, 640), RefSinc(640, 480)
    y = np.stack([up.resample(fr) for fr in x])
    z = np.stack([down.resample(fr) for fr in y])
    path = os.path.join(ROOT, "tests", "golden", "sinc_golden.npz")
    np.savez_compressed(path, x48=x, y64=y, z48=z)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
