"""CPU suite: the register layouts and XOR swizzles of the BlockThresholding N = 1024 kernel
(audiosignalprocess_amd/csrc/bt_layout.h, shared by bt_kernels8.hip and the host table in bt_api.hip).
Compiled here with g++ (the header is plain constexpr C++): every exchange must be a bijection onto the
512 slots of a wave's row, its 8-byte writes and reads free of LDS bank conflicts under the gfx950 rules
(MI355X_MICROARCH.md: ds_write_b64 = 16-lane groups over 32 dword banks, ds_read_b64 = 32-lane groups
over 64 dword banks), and the slot of (lane, register) must split into a lane term XOR a register term
whose bits above the swizzled five add (what the kernel's addressing relies on)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include "bt_layout.h"
using namespace aspbt;
int main() {
  const Lay* lay[4] = {&LA, &LB, &LC, &LD};
  const Swz* sw[3] = {&S1, &S2, &S3};
  for (int x = 0; x < 3; ++x)
    for (int side = 0; side < 2; ++side)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const Lay& L = *lay[x + side];
          const int p = pos_lane(L, lane) | pos_reg(L, j);
          std::printf("%d %d %d %d %d %d %d %d\n", x, side, lane, j, p, swz(*sw[x], p), swz(*sw[x], pos_lane(L, lane)),
                      swz(*sw[x], pos_reg(L, j)));
        }
  return 0;
}
"""


def _table(tmp_path):
    src = tmp_path / "lay.cpp"
    src.write_text(SRC)
    exe = tmp_path / "lay"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "audiosignalprocess_amd", "csrc"),
                    str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    return np.array([[int(v) for v in line.split()] for line in out.splitlines()])


def test_exchanges_are_bijections_without_bank_conflicts(tmp_path):
    t = _table(tmp_path)
    assert t.shape == (3 * 2 * 64 * 8, 8)
    for x in range(3):
        for side in range(2):
            m = t[(t[:, 0] == x) & (t[:, 1] == side)]
            lane, j, p, slot, lterm, rterm = m[:, 2], m[:, 3], m[:, 4], m[:, 5], m[:, 6], m[:, 7]
            assert sorted(p) == list(range(512)) and sorted(slot) == list(range(512))
            # the kernel's addressing: slot = lane term ^ register term; the register term's bits >= 32 are
            # disjoint from the lane term's, so they can be added as an immediate offset
            assert np.array_equal(slot, lterm ^ rterm)
            assert not np.any((lterm & ~31) & (rterm & ~31))
            assert np.array_equal(slot, (lterm ^ (rterm & 31)) + (rterm & ~31))
            group, banks = (16, 16) if side == 0 else (32, 32)   # 8-byte slots per bank cycle
            for jj in range(8):
                for g0 in range(0, 64, group):
                    sel = (j == jj) & (lane >= g0) & (lane < g0 + group)
                    assert len(set(slot[sel] % banks)) == group, (x, side, jj, g0)


def test_layout_chain_matches_kiss_fft_stages(tmp_path):
    """Stage inputs: layout A holds position bits 0..2 in registers (radix 2 + radix 4 with m = 2), B bits
    3, 4 (m = 8), C bits 5, 6 (m = 32), D bits 7, 8 (m = 128) -- each on register bits 1 and 2 -- and the
    natural order p = lane + 64 j at the end."""
    t = _table(tmp_path)
    for x, side, bits in ((0, 0, (1, 2)), (0, 1, (3, 4)), (1, 1, (5, 6)), (2, 1, (7, 8))):
        m = t[(t[:, 0] == x) & (t[:, 1] == side) & (t[:, 2] == 5)]
        p = {int(r[3]): int(r[4]) for r in m}
        assert p[2] - p[0] == 1 << bits[0] and p[4] - p[0] == 1 << bits[1]
    d = t[(t[:, 0] == 2) & (t[:, 1] == 1)]
    assert np.array_equal(d[:, 4], d[:, 2] + 64 * d[:, 3])
