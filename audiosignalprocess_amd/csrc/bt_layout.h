// bt_layout.h -- device-side layout shared by bt_kernels.hip, bt_kernels8.hip and bt_api.hip.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BT_HD __host__ __device__
#else
#define BT_HD
#endif

namespace aspbt {

// per-stream carried state: [inbuf tail (<= 512)] [outbuf tail (<= 512)]
constexpr int kOffInTail = 0;
constexpr int kOffOutTail = 512;
constexpr int kStateFloats = 1024;

// constants of one dyadic segmentation (T, F): audioDenoiseBlockTreshold.c:383-390, 431
struct BtSeg {
  float size_blk;  // TT * FF
  float temp;      // lambda^2 size^2 - 2 lambda size (size - 2)
  float thr;       // lambda * size
  float two_size;  // 2 * size
  float a_const;   // (float)(lambda TT FF sigma_h^2 N)
};

struct BtSize {
  float norm;      // sqrt(2) / (sqrt(N) sigma_h)         (.c:365)
  float dc_const;  // Lambda_pi L_pi sigma_h^2 N            (.c:503, 523)
  float wiener_c;  // N sigma_h^2                           (.c:481)
  float pad;
  BtSeg seg[3][5];
};

struct BtTables {
  float hann256[256], hann1024[1024];            // make_hanning_window (.c:70-77)
  float tw256_f[2 * 128], tw256_i[2 * 128];      // kiss_fft twiddles, nfft = 128 (kiss_fft.c:357-363)
  float tw1024_f[2 * 512], tw1024_i[2 * 512];    // nfft = 512
  float sup256_f[2 * 64], sup256_i[2 * 64];      // kiss_fftr super twiddles (kiss_fftr.c:57-63)
  float sup1024_f[2 * 256], sup1024_i[2 * 256];
  BtSize s256, s1024;
  // bt_kernels8.hip: byte offset (8 * swizzled slot) of a lane's part of the exchange addresses,
  // [2 x]: source layout of exchange x + 1, [2 x + 1]: its destination layout
  uint16_t xterm[6][64];
};

// Any other even window (4 .. 2048 samples): the tables of one batch, handed to
// bt_macroblock_any_kernel by value.  kiss_fft's plan (kf_factor, kiss_fft.c:308-330) as
// (radix, remaining length) pairs, outermost first; `perm[n]` is where input n of the
// N/2-point transform lands after kf_work's recursive decimation (kiss_fft.c:237-302).
constexpr int kAnyMaxWin = 2048;        // 8 x (win / 2 + 1) coefficients twice over must fit the 160 KB of LDS
constexpr int kAnyMaxRadix = 32;        // largest prime factor the generic butterfly's scratch holds
constexpr int kAnyStateFloats = 2048;   // per-stream carried state of these batches: [inbuf tail (<= 1024)] [outbuf tail]
constexpr int kAnyOffOutTail = 1024;
struct BtAnyTables {
  int n, nc, ncol, nfac;
  int fac[2 * 16];
  BtSize P;
  const float* hann;     // [n]
  const float* tw_f;     // [nc] complex, forward
  const float* tw_i;     // [nc] complex, inverse
  const float* sup_f;    // [nc / 2] complex
  const float* sup_i;
  const uint16_t* perm;  // [nc]
  // the device copy of this struct (at the head of the device block the pointers above point into): the
  // kernels take it by address -- passed by value the whole struct was copied into SGPRs at kernel entry and
  // spilled from there (273 spilled SGPRs)
  const BtAnyTables* self;
};

// ---------------------------------------------------------------- register layouts of the FFT stages of bt_kernels8.hip
// A position p (9 bits) of the in-place kiss_fft work array lives in lane `lane`, register j of a
// layout: reg[b] / lane[b] name the position bit held by register-index bit b / lane bit b.
struct Lay {
  int reg[3];
  int lane[6];
};
constexpr Lay LA = {{0, 1, 2}, {7, 8, 5, 6, 3, 4}};  // after the loads: lane = k0 + 4 k1 + 16 k2 of n
constexpr Lay LB = {{8, 3, 4}, {0, 1, 2, 5, 6, 7}};  // radix-4 stage m = 8 (position bits 3, 4 in registers)
constexpr Lay LC = {{8, 5, 6}, {0, 1, 2, 3, 4, 7}};  // m = 32
constexpr Lay LD = {{6, 7, 8}, {0, 1, 2, 3, 4, 5}};  // m = 128; natural order: p = lane + 64 j
// XOR swizzles of the three exchanges (index bit b = parity(p & rows[b]) for b < 5, bits 5..8 kept):
// found by search so that the ds_write_b64 of the source layout (16-lane groups, 16 8-byte banks) and
// the ds_read_b64 of the destination layout (32-lane groups, 32 8-byte banks) are both conflict-free.
struct Swz {
  int rows[5];
};
constexpr Swz S1 = {{257, 322, 396, 40, 80}};
constexpr Swz S2 = {{129, 34, 4, 56, 16}};
constexpr Swz S3 = {{17, 66, 12, 264, 16}};

BT_HD constexpr int pos_reg(const Lay& L, int j) {
  int p = 0;
  for (int b = 0; b < 3; ++b) p |= ((j >> b) & 1) << L.reg[b];
  return p;
}
BT_HD constexpr int pos_lane(const Lay& L, int lane) {
  int p = 0;
  for (int b = 0; b < 6; ++b) p |= ((lane >> b) & 1) << L.lane[b];
  return p;
}
BT_HD constexpr int parity9(int x) {
  x ^= x >> 8;
  x ^= x >> 4;
  x ^= x >> 2;
  x ^= x >> 1;
  return x & 1;
}
BT_HD constexpr int swz(const Swz& S, int p) {
  int r = p & ~31;
  for (int b = 0; b < 5; ++b) r |= parity9(p & S.rows[b]) << b;
  return r;
}

}  // namespace aspbt
