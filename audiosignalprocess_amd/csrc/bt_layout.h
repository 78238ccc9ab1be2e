// bt_layout.h -- device-side layout shared by bt_kernels.hip and bt_api.hip.
#pragma once
#include <stdint.h>

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
};

}  // namespace aspbt
