/*
 * bt_oracle.h -- CPU restatement of Denoise/BlockThresholding (+ the kiss_fft
 * subset it uses).  TEST INFRASTRUCTURE ONLY: only tests/, smoke() and
 * bench.py's cpu_baseline leg may build, load or call it.
 *
 * Parity status: UNPINNED for the FFT internals (the reference's
 * common/kiss_fft/_kiss_fft_guts.h is missing, so the reference cannot be
 * compiled here and it ships no expected outputs); see bt_oracle.c.
 */
#ifndef ASP_BT_ORACLE_H_
#define ASP_BT_ORACLE_H_

#include "asp_bt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct BtOracle BtOracle;

BtOracle* bt_oracle_create(int win_size); /* 256 or 1024; NULL otherwise */
void bt_oracle_free(BtOracle* h);
void bt_oracle_reset(BtOracle* h);
int bt_oracle_win(const BtOracle* h);
/* hop-level protocol of the reference (audioDenoiseBlockTreshold.c:541-672) */
int bt_oracle_denoise_float(BtOracle* h, const float* in, int in_len);
int bt_oracle_output_float(BtOracle* h, float* out, int out_len);
int bt_oracle_flush_float(BtOracle* h, float* out, int out_len);
/* one macroblock = 8 hops: in/out [8 * win/2]; seg_out (may be NULL) gets
 * (seg_time, seg_freq) per macro-column */
void bt_oracle_macroblock(BtOracle* h, const float* in, float* out, int* seg_out);
void bt_oracle_export(const BtOracle* h, AspBtState* s);
void bt_oracle_import(BtOracle* h, const AspBtState* s);
float bt_oracle_s16_to_float(int16_t v);
int16_t bt_oracle_float_to_s16(float v);
void bt_oracle_kiss_fftr(BtOracle* h, const float* timedata, float* freq);
void bt_oracle_kiss_fftri(BtOracle* h, const float* freq, float* timedata);
const float* bt_oracle_hann(const BtOracle* h);

#ifdef __cplusplus
}
#endif
#endif
