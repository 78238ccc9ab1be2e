#!/usr/bin/env python3
"""Generates tests/golden/ns_hb_golden.npz from the REFERENCE NS build (oracle/_ref/libns_ref.so):
WebRtcNs_AnalyzeCore(low) + WebRtcNs_ProcessCore(2 bands) at 32 kHz, policy 2.  Build container only.
  low_i16, high_i16 [F][S][160] int16-valued band samples (what the QMF split produces),
  out_low, out_high [F][S][160] float32 reference outputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import RefNs, have_ref  # noqa: E402
from tests.test_ns_oracle import _band_frames  # noqa: E402

S, F = 1, 540


def main():
    assert have_ref(), "build oracle/_ref first (make -C oracle)"
    low, high = _band_frames(S, F, 1)
    low = np.clip(np.rint(low), -32768, 32767).astype(np.int16)
    high = np.clip(np.rint(high[:, 0]), -32768, 32767).astype(np.int16)
    ref = RefNs(S, policy=2, fs=32000)
    ol, oh = ref.run_bands(low.astype(np.float32), high.astype(np.float32)[:, None])
    path = os.path.join(ROOT, "tests", "golden", "ns_hb_golden.npz")
    np.savez_compressed(path, low_i16=low, high_i16=high, out_low=ol, out_high=oh[:, 0])
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
