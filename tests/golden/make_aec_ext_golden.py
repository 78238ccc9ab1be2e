#!/usr/bin/env python3
"""Generates tests/golden/aec_ext_golden.npz from the REFERENCE AEC build (oracle/_ref/libaec_ref.so)
with the extended filter switched on -- WebRtcAec_enable_delay_correction(WebRtcAec_aec_core(h), 1),
aec_core.c:1876-1881: 32 partitions, ProcessExtended / EstBufDelayExtended.  Build-container only.

  far_i16, near_i16 [F][S][160] int16   synthetic far / near-end PCM (SURVEY 8(d) generator)
  out_f32           [F][S][160] float32 WebRtcAec_BufferFarend + WebRtcAec_Process output
  delay_ms                              the msInSndCardBuf every call reports
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiosignalprocess_amd.synth import aec_frames  # noqa: E402
from tests.oracle_lib import RefAec, have_aec_ref  # noqa: E402

S, F, DELAY = 4, 500, 40


def main():
    assert have_aec_ref(), "build oracle/_ref first (make -C oracle)"
    far, near = aec_frames(S, F)
    far = np.clip(np.rint(far), -32768, 32767).astype(np.int16)
    near = np.clip(np.rint(near), -32768, 32767).astype(np.int16)
    out = np.empty((F, S, 160), np.float32)
    for s in range(S):
        r = RefAec()
        r.enable_delay_correction(1)
        out[:, s] = r.run(far[:, s].astype(np.float32), near[:, s].astype(np.float32), DELAY)
    path = os.path.join(ROOT, "tests", "golden", "aec_ext_golden.npz")
    np.savez_compressed(path, far_i16=far, near_i16=near, out_f32=out, delay_ms=np.int32(DELAY))
    seg = slice(160, 290)
    e_in = (near[seg].astype(np.float64) ** 2).mean()
    e_out = (out[seg].astype(np.float64) ** 2).mean()
    print("wrote", path, os.path.getsize(path), "bytes; ERLE %.1f dB" % (10 * np.log10(e_in / e_out)))


if __name__ == "__main__":
    main()
