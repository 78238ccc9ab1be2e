#!/usr/bin/env python3
"""Generates tests/golden/split_golden.npz from the REFERENCE SplittingFilter build
(oracle/_ref/libsplit_ref.so).  Build container only.
  x48 [F][480] int16 -> bands [F][3][160] (Analysis) -> merged [F][480] (Synthesis of those bands)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import RefSplit, have_split_ref  # noqa: E402
from tests.test_sinc_oracle import sinc_inputs  # noqa: E402


def main():
    assert have_split_ref(), "build oracle/_ref first (make -C oracle)"
    x = sinc_inputs(30, 480, seed=29)
    ref = RefSplit(3)
    bands, merged = [], []
    for fr in x:
        b = ref.analysis(fr)
        bands.append(b)
        merged.append(ref.synthesis(b))
    path = os.path.join(ROOT, "tests", "golden", "split_golden.npz")
    np.savez_compressed(path, x48=x, bands=np.stack(bands), merged=np.stack(merged))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
