#!/usr/bin/env python3
"""Generates tests/golden/qmf_golden.npz from the REFERENCE QMF build (oracle/_ref/libspl_ref.so:
the reference's splitting_filter_c.c compiled in place).  Build-container only.
  x [F][320] int16 input frames (32 kHz, 10 ms), low / high [F][160] analysis outputs,
  merged [F][320] synthesis of those bands, state [24] the four filter states at the end."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import RefQmf, have_spl_ref  # noqa: E402
from tests.test_qmf_oracle import qmf_inputs  # noqa: E402


def main():
    assert have_spl_ref(), "build oracle/_ref first (make -C oracle)"
    x = qmf_inputs()
    ref = RefQmf()
    low, high, merged = [], [], []
    for fr in x:
        lo, hi = ref.analysis(fr)
        low.append(lo)
        high.append(hi)
        merged.append(ref.synthesis(lo, hi))
    path = os.path.join(ROOT, "tests", "golden", "qmf_golden.npz")
    np.savez_compressed(path, x=x, low=np.stack(low), high=np.stack(high), merged=np.stack(merged),
                        state=ref.state())
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
