/*
 * asp_bt.h -- C-ABI of the MI355X batched time-frequency block-thresholding
 * denoiser (Yu/Mallat/Bacry), the reference's Denoise/BlockThresholding.
 *
 * Layer 1: the reference's own per-stream API, signature-identical
 * (Denoise/BlockThresholding/src/audioDenoiseBlockTreshold.h:46-74).  The
 * header declares blockThreshold_flush_float with a float* buffer (.h:66)
 * while the definition takes int16_t* (.c:648); the header form is exported.
 * Layer 2: AspBtBatch_*, N independent stream-channels per call, one
 * 8-hop macroblock per stream-channel per call.
 *
 * Window sizes: the reference derives win_size = fs/1000*time_win (.c:91) and
 * kiss_fft factors any length.  This build: every even window of 4 .. 2048
 * samples whose half has no prime factor above 32 (20 ms at 16 kHz = 320,
 * 50 ms = 800, 10 / 20 / 40 ms at 48 kHz = 480 / 960 / 1920, ...) through kiss_fft's
 * mixed-radix plan (radix 4, 2, 3, 5 and the generic butterfly), with tuned
 * kernels for 256 and 1024 (BASELINE configs 1 and 3); anything else
 * (longer than 2048 samples, a larger prime) returns MARS_ERROR_PARAMS.
 */
#ifndef ASP_BT_H_
#define ASP_BT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* audioDenoiseBlockTreshold.h:7-11 */
#define MARS_OK 0x00
#define MARS_ERROR_MEMORY 0x01
#define MARS_ERROR_PARAMS 0x02
#define MARS_NEED_MORE_SAMPLES 0x10
#define MARS_CAN_OUTPUT 0x20

#define ASP_BT_NBLK_TIME 8  /* max_nblk_time, .c:107 */
#define ASP_BT_NBLK_FREQ 16 /* max_nblk_freq, .c:108 */

/* ---------------------------------------------------------------- layer 1 */
typedef struct MarsBlockThreshold MarsBlockThreshold_t;

MarsBlockThreshold_t* blockThreshold_init(int32_t time_win, int32_t fs, int32_t* err); /* .h:46 */
int32_t blockThreshold_reset(MarsBlockThreshold_t* handle);                            /* .h:48 */
int32_t blockThreshold_denoise_int16(MarsBlockThreshold_t* handle, int16_t* in, int32_t in_len);
int32_t blockThreshold_denoise_float(MarsBlockThreshold_t* handle, float* in, int32_t in_len);
int32_t blockThreshold_output_int16(MarsBlockThreshold_t* handle, int16_t* out, int32_t out_len);
int32_t blockThreshold_output_float(MarsBlockThreshold_t* handle, float* out, int32_t out_len);
int32_t blockThreshold_flush_int16(MarsBlockThreshold_t* handle, int16_t* out, int32_t out_len);
int32_t blockThreshold_flush_float(MarsBlockThreshold_t* handle, float* out, int32_t out_len);
void blockThreshold_free(MarsBlockThreshold_t* handle);
int32_t blockThreshold_max_output(const MarsBlockThreshold_t* handle);       /* macro_size    */
int32_t blockThreshold_samples_per_time(const MarsBlockThreshold_t* handle); /* half_win_size */

/* ---------------------------------------------------------------- layer 2 */
typedef struct AspBtBatch AspBtBatch;

/* Per-stream carried state between macroblocks (checkpoint / parity tests):
 * the second half of the analysis buffer and the overlap-add tail. */
typedef struct AspBtState {
  int32_t win_size;
  float inbuf_tail[1024]; /* inbuf[half..win), first half_win entries used */
  float out_tail[1024];   /* outbuf[macro..macro+half)                     */
} AspBtState;

int AspBtBatch_Create(AspBtBatch** out, int num_streams, int win_size, int device);
int AspBtBatch_Free(AspBtBatch* b);
int AspBtBatch_Reset(AspBtBatch* b);
/* blockThreshold_reset of ONE stream-channel of a running batch: its two tails are cleared, the others untouched. */
int AspBtBatch_ResetStream(AspBtBatch* b, int stream);
/* `nblocks` consecutive macroblocks (8 hops each) of every stream-channel in one call: in / out
 * [nblocks][num_streams][macro_size] float32.  With 256- and 1024-sample windows the macroblocks of up to 64 steps
 * go into ONE launch (the hand-off build: a stream-channel's input tail is read from the previous macroblock's
 * input, its overlap-add tail is handed on through memory behind a per-stream counter); results are those of
 * nblocks AspBtBatch_Denoise calls, bit for bit.  AspBtBatch_SetFlow: -1 default (on; ASP_BT_FLOW=0 in the
 * environment turns the default off), 0 off (one launch per macroblock), 1 on. */
int AspBtBatch_DenoiseBlocks(AspBtBatch* b, const float* in, float* out, int nblocks, int mem);
int AspBtBatch_SetFlow(AspBtBatch* b, int mode);
int AspBtBatch_num_streams(const AspBtBatch* b);
int AspBtBatch_macro_size(const AspBtBatch* b); /* 8 * win_size / 2 samples per call */
/* One macroblock per stream: in/out [num_streams][macro_size] float in [-1, 1]
 * units (the reference's float path, .c:541-575); out lags in by half a
 * window, as in the reference.  mem: 0 host, 1 device (asp_ns.h ASP_MEM_*). */
int AspBtBatch_Denoise(AspBtBatch* b, const float* in, float* out, int mem);
/* blockThreshold_flush_float for every stream with `hops` (0..7) pending hops:
 * in [num_streams][hops*half], out [num_streams][hops*half]; no thresholding
 * is applied to a partial macroblock (.c:648-672).  As in the reference the call
 * consumes the overlap tail (a second flush, or the macroblock the pending hops
 * later complete, starts from a cleared tail) and leaves the input history
 * alone (the pending hops stay pending and are fed again with their macroblock). */
int AspBtBatch_Flush(AspBtBatch* b, const float* in, int hops, float* out, int mem);
int AspBtBatch_ExportState(AspBtBatch* b, int stream, AspBtState* out);
int AspBtBatch_ImportState(AspBtBatch* b, int stream, const AspBtState* in);
int AspBtBatch_Synchronize(AspBtBatch* b);
int AspBtBatch_TimedSteps(AspBtBatch* b, const float* in, float* out, int blocks_in_ring,
                          int steps, float* elapsed_ms);

/* kiss_fftr / kiss_fftri seam for the parity tests (common/kiss_fft/kiss_fftr.c:67-159):
 * forward: time [count][n] -> freq [count][n/2+1] interleaved {r,i};
 * inverse: freq -> time, unscaled.  n in {256, 1024}. */
int AspBt_kiss_fftr_batch(const float* timedata, float* freqdata, int n, int count, int device);
int AspBt_kiss_fftri_batch(const float* freqdata, float* timedata, int n, int count, int device);

#ifdef __cplusplus
}
#endif
#endif /* ASP_BT_H_ */
