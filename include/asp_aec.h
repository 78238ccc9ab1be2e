/*
 * asp_aec.h -- C-ABI of the MI355X batched acoustic echo canceller: the reference's
 * WebRTC AEC (partitioned-block frequency-domain adaptive filter + non-linear
 * processor), WebRtc_AMP_Port/webrtc/modules/audio_processing/aec/.
 *
 * Layer 1: the reference's per-stream API, signature-identical
 * (aec/include/echo_cancellation.h:79-247).  A handle is a batch of one stream.
 * Layer 2: AspAecBatch_*, N independent streams fed in lock-step: the integer control
 * plane of the reference (start-up phase, system-delay bookkeeping, ring read/write
 * positions: echo_cancellation.c:594-742,816-867, aec_core.c:1618-1778) depends only on
 * the call sequence and the reported delays, so one host copy serves every stream of a
 * batch; the per-stream float state and the four ring buffers live in HBM.
 *
 * (The delay-agnostic mode is the exception: there every stream steers its far-end read pointer by
 * its own delay estimate, and that part of the control plane runs per stream on the device.)
 *
 * Built configuration = what test_aec_module.cpp:60-88 runs -- 8 or 16 kHz (one band), 12 partitions,
 * reported-delay mode -- plus 32 kHz (two bands) and the reference's optional modes: echo metrics
 * (ERL / ERLE / A_NLP), the extended filter (32 partitions), delay logging with GetDelayMetrics, the
 * delay-agnostic mode (reported delays off) and skew compensation.  48 kHz is refused (the reference's
 * own init breaks there) with the reference's error codes.
 */
#ifndef ASP_AEC_H_
#define ASP_AEC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A translation unit that already included the reference's own echo_cancellation.h keeps its
 * (identical) layer-1 declarations. */
#ifndef WEBRTC_MODULES_AUDIO_PROCESSING_AEC_INCLUDE_ECHO_CANCELLATION_H_
/* echo_cancellation.h:17-24 */
#define AEC_UNSPECIFIED_ERROR 12000
#define AEC_UNSUPPORTED_FUNCTION_ERROR 12001
#define AEC_UNINITIALIZED_ERROR 12002
#define AEC_NULL_POINTER_ERROR 12003
#define AEC_BAD_PARAMETER_ERROR 12004
#define AEC_BAD_PARAMETER_WARNING 12050

/* ---------------------------------------------------------------- layer 1 */
enum { kAecNlpConservative = 0, kAecNlpModerate, kAecNlpAggressive }; /* echo_cancellation.h:26-30 */
enum { kAecFalse = 0, kAecTrue };                                      /* echo_cancellation.h:32-35 */

typedef struct {
  int16_t nlpMode;      /* default kAecNlpModerate */
  int16_t skewMode;     /* default kAecFalse       */
  int16_t metricsMode;  /* default kAecFalse       */
  int delay_logging;    /* default kAecFalse       */
} AecConfig;            /* echo_cancellation.h:37-43 */

typedef struct {
  int instant;
  int average;
  int max;
  int min;
} AecLevel; /* echo_cancellation.h:45-50 */

typedef struct {
  AecLevel rerl;
  AecLevel erl;
  AecLevel erle;
  AecLevel aNlp;
} AecMetrics; /* echo_cancellation.h:52-57 */

struct AecCore;

int32_t WebRtcAec_Create(void** aecInst);                                       /* .h:79  */
int32_t WebRtcAec_Free(void* aecInst);                                          /* .h:93  */
int32_t WebRtcAec_Init(void* aecInst, int32_t sampFreq, int32_t scSampFreq);    /* .h:109 */
int32_t WebRtcAec_BufferFarend(void* aecInst, const float* farend, int16_t nrOfSamples); /* .h:126 */
int32_t WebRtcAec_Process(void* aecInst, const float* const* nearend, int num_bands,
                          float* const* out, int16_t nrOfSamples, int16_t msInSndCardBuf,
                          int32_t skew);                                        /* .h:153 */
int WebRtcAec_set_config(void* handle, AecConfig config);                       /* .h:175 */
int WebRtcAec_get_echo_status(void* handle, int* status);                       /* .h:191 */
/* GetMetrics: echo_cancellation.c:456-548 (the values stay at kOffsetLevel = -100 until
 * set_config enables metricsMode).  GetDelayMetrics: echo_cancellation.c:550-571 (set_config with
 * delay_logging = kAecTrue; AEC_UNSUPPORTED_FUNCTION_ERROR while logging is off). */
int WebRtcAec_GetMetrics(void* handle, AecMetrics* metrics);                    /* .h:207 */
int WebRtcAec_GetDelayMetrics(void* handle, int* median, int* std);             /* .h:224 */
int32_t WebRtcAec_get_error_code(void* aecInst);                                /* .h:237 */
/* The reference returns its internal AecCore; here it is an opaque token that is only
 * valid as an argument of the layer-1 functions of this library. */
struct AecCore* WebRtcAec_aec_core(void* handle);                               /* .h:247 */
/* The extended filter (32 partitions, ProcessExtended delay handling): aec_core.h:106-113,
 * aec_core.c:1876-1885.  As in the reference, Init switches it off again. */
void WebRtcAec_enable_delay_correction(struct AecCore* self, int enable);
int WebRtcAec_delay_correction_enabled(struct AecCore* self);
/* The buffered far-end delay in samples (aec_core.h:115-122, aec_core.c:1886-1894). */
int WebRtcAec_system_delay(struct AecCore* self);
void WebRtcAec_SetSystemDelay(struct AecCore* self, int delay);
/* Reported delays off = the delay-agnostic mode (aec_core.h:97-104, aec_core.c:1868-1874). */
void WebRtcAec_enable_reported_delay(struct AecCore* self, int enable);
int WebRtcAec_reported_delay_enabled(struct AecCore* self);
#endif /* reference header not included */

#define ASP_AEC_PART_LEN 64    /* aec_core.h:21 */
#define ASP_AEC_PART_LEN1 65   /* aec_core.h:22 */
#define ASP_AEC_PART_LEN2 128  /* aec_core.h:23 */
#define ASP_AEC_FRAME_LEN 80   /* aec_core.h:20 */
#define ASP_AEC_PARTITIONS 12      /* kNormalNumPartitions, aec_core_internal.h:25 */
#define ASP_AEC_PARTITIONS_MAX 32  /* kExtendedNumPartitions, aec_core_internal.h:23: the length of the reference's
                                    * arrays whatever the filter length (extended filter: all 32 live) */
#define ASP_AEC_FAR_SLOTS 250  /* kBufSizePartitions, aec_core.c:37 */

/* ---------------------------------------------------------------- layer 2 */
typedef struct AspAecBatch AspAecBatch;

/* Per-stream float state between blocks, canonical field names of AecCore
 * (aec_core_internal.h:52-164).  Checkpoint unit and parity-test unit. */
typedef struct AspAecState {
  float dBuf[128];
  float eBuf[128];
  float xPow[65];
  float dPow[65];
  float dMinPow[65];
  float dInitMinPow[65];
  float xfBuf[2][ASP_AEC_PARTITIONS_MAX * 65];  /* partition p at [.][p * 65 ..] */
  float wfBuf[2][ASP_AEC_PARTITIONS_MAX * 65];
  float sde[65][2];
  float sxd[65][2];
  float xfwBuf[ASP_AEC_PARTITIONS_MAX * 65][2];
  float sx[65];
  float sd[65];
  float se[65];
  float outBuf[64];
  float hNlFbMin, hNlFbLocalMin, hNlXdAvgMin;
  float overDrive, overDriveSm;
  int32_t hNlNewMin, hNlMinCtr;
  int32_t delayIdx, stNearState, echoState, divergeState;
  int32_t xfBufBlockPos, noiseEstCtr, delayEstCtr;
  uint32_t seed;
  float dBufH[128]; /* first high band (32 kHz), aec_core_internal.h:67 */
} AspAecState;

/* Echo metrics state of one stream (metricsMode = kAecTrue): PowerLevel x 4 and Stats x 4 of
 * AecCore (aec_core_internal.h:39-47,126-135, aec_core.h:30-40) and stateCounter. */
typedef struct AspAecPowerLevel {
  float sfrsum;
  int32_t sfrcounter;
  float framelevel, frsum;
  int32_t frcounter;
  float minlevel, averagelevel;
} AspAecPowerLevel;
typedef struct AspAecStats {
  float instant, average, min, max, sum, hisum, himean;
  int32_t counter, hicounter;
} AspAecStats;
typedef struct AspAecMetricsState {
  AspAecPowerLevel farlevel, nearlevel, linoutlevel, nlpoutlevel;
  AspAecStats erl, erle, aNlp, rerl;
  int32_t stateCounter;
} AspAecMetricsState;

/* Delay estimation of one stream (set_config with delay_logging = kAecTrue; also what the
 * delay-agnostic mode steers by): the binary-spectrum delay estimator of utility/delay_estimator.c /
 * delay_estimator_wrapper.c (float path) with the sizes aec_core.c:1356-1377 creates it with, and the
 * AecCore fields around it (aec_core_internal.h:129-142). */
#define ASP_AEC_DELAY_HISTORY 125   /* kHistorySizeBlocks, aec_core_internal.h:34 */
#define ASP_AEC_DELAY_LOOKAHEAD 15  /* kLookaheadBlocks, aec_core_internal.h:30   */
typedef struct AspAecDelayState {
  /* far end: DelayEstimatorFarend + BinaryDelayEstimatorFarend */
  float mean_far_spectrum[65];
  int32_t far_spectrum_initialized;
  uint32_t binary_far_history[ASP_AEC_DELAY_HISTORY];
  int32_t far_bit_counts[ASP_AEC_DELAY_HISTORY];
  /* near end: DelayEstimator + BinaryDelayEstimator */
  float mean_near_spectrum[65];
  int32_t near_spectrum_initialized;
  uint32_t binary_near_history[ASP_AEC_DELAY_HISTORY + 1]; /* max_lookahead = kHistorySizeBlocks, aec_core.c:1363-1366 */
  int32_t mean_bit_counts[ASP_AEC_DELAY_HISTORY + 1];
  int32_t bit_counts[ASP_AEC_DELAY_HISTORY];
  float histogram[ASP_AEC_DELAY_HISTORY + 1];
  int32_t minimum_probability, last_delay_probability, last_delay;
  int32_t last_candidate_delay, compare_delay, candidate_hits;
  float last_delay_histogram;
  int32_t lookahead, allowed_offset;
  /* AecCore: the logging histogram and the signal-based correction of the delay-agnostic mode */
  int32_t delay_histogram[ASP_AEC_DELAY_HISTORY];
  int32_t previous_delay, delay_correction_count, shift_offset;
  float delay_quality_threshold;
  /* the stream's own far-buffer read side and system delay (identical for all streams of a batch until
   * the delay-agnostic mode moves them apart) */
  int32_t far_read, far_write, far_wrap, system_delay;
} AspAecDelayState;

/* Integer control plane shared by all streams of a batch (one Aec + the integer part of
 * AecCore), exposed for the parity tests. */
typedef struct AspAecControl {
  int32_t startup_phase, checkBuffSize, bufSizeStart, knownDelay, filtDelay;
  int32_t timeForDelayChange, lastDelayDiff, counter, sum, firstVal, checkBufSizeCtr;
  int32_t system_delay, core_knownDelay;
  int32_t far_read, far_write, far_wrap;       /* far_buf / far_buf_windowed (lock-step) */
  int32_t pre_read, pre_write, pre_wrap;       /* far_pre_buf                            */
  int32_t near_read, near_write, near_wrap;    /* nearFrBuf                              */
  int32_t out_read, out_write, out_wrap;       /* outFrBuf                               */
  int32_t blocks_processed;
} AspAecControl;

int AspAecBatch_Create(AspAecBatch** out, int num_streams, int device);
int AspAecBatch_Free(AspAecBatch* b);
/* The integer control plane alone: no device is touched and nothing is launched; Init / set_config /
 * BufferFarend / Process / GetControl work (buffers are only checked for NULL), everything that
 * touches data refuses the handle.  For checking the host logic on a machine without a GPU. */
int AspAecBatch_CreateControlOnly(AspAecBatch** out, int num_streams);
/* WebRtcAec_Init for every stream; sampFreq 8000 or 16000. */
int AspAecBatch_Init(AspAecBatch* b, int32_t sampFreq, int32_t scSampFreq);
int AspAecBatch_set_config(AspAecBatch* b, AecConfig config);
int AspAecBatch_num_streams(const AspAecBatch* b);
/* WebRtcAec_BufferFarend for every stream: farend [num_streams][nrOfSamples] float-S16,
 * nrOfSamples 80 or 160.  mem: 0 host, 1 device (asp_ns.h ASP_MEM_*). */
int AspAecBatch_BufferFarend(AspAecBatch* b, const float* farend, int nrOfSamples, int mem);
/* WebRtcAec_Process for every stream: nearend / out [num_streams][nrOfSamples]; in-place
 * allowed.  Returns 0 / -1 like the reference (-1 with AEC_BAD_PARAMETER_WARNING still
 * processes, echo_cancellation.c:367-375). */
int AspAecBatch_Process(AspAecBatch* b, const float* nearend, float* out, int nrOfSamples,
                        int msInSndCardBuf, int32_t skew, int mem);
/* 32 kHz (AspAecBatch_Init(b, 32000, ...)): WebRtcAec_Process with num_bands = 2: the 0-8 kHz band
 * is cancelled as above, the high band is delayed alongside and scaled by the average NLP gain
 * of the upper half of the low band plus comfort noise (aec_core.c:501-545, 1032-1067).
 * near_high / out_high [num_streams][nrOfSamples].  (48 kHz: the reference's own InitAec breaks
 * there -- (short)48000 / 16000 = -1, aec_core.c:1541-1543 -- and is refused.) */
int AspAecBatch_ProcessBands(AspAecBatch* b, const float* near_low, const float* near_high,
                             float* out_low, float* out_high, int nrOfSamples, int msInSndCardBuf,
                             int32_t skew, int mem);
int AspAecBatch_num_bands(const AspAecBatch* b);

/* ---- per-stream control.  The reference takes the reported delay, Init and the call pattern per handle
 * (WebRtcAec_Process(aecInst, ..., msInSndCardBuf, skew), echo_cancellation.c:341-347; WebRtcAec_Init, :196-276).  A
 * batch starts in lock-step: one control plane (ring positions, start-up phase, EstBufDelay's filter, knownDelay) for
 * all its streams, the far-end work fused into the Process launch.  The first AspAecBatch_ProcessV or
 * AspAecBatch_InitStream call gives EVERY stream its own control plane (kept on the host, run once per stream and
 * call; the kernels read one descriptor per stream from device memory), and the uniform-argument entry points
 * (BufferFarend, Process, Run, TimedSteps) keep working on top of it.  AspAecBatch_Init returns to lock-step.
 * Covered: one band (8 / 16 kHz), the normal and the extended filter, reported delays.  Not covered (ASP_ERR_STATE):
 * 32 kHz, delay logging, the delay-agnostic mode, skew compensation, metrics.
 *   ProcessV     msInSndCardBuf[num_streams]: each stream's reported delay; skew may be null (ignored);
 *                status[num_streams] (may be null): each stream's reference return value (0, or -1 for a delay
 *                outside 0 .. 500 ms, which still processes, echo_cancellation.c:367-375); returns -1 if any is.
 *   InitStream   WebRtcAec_Init of ONE stream at the batch's sample rates: its control plane, state block and
 *                far-ring slots back to their initial values (start-up phase included); the others untouched.
 *   GetControlStream   AspAecBatch_GetControl of one stream. */
int AspAecBatch_ProcessV(AspAecBatch* b, const float* nearend, float* out, int nrOfSamples,
                         const int16_t* msInSndCardBuf, const int32_t* skew, int32_t* status, int mem);
int AspAecBatch_InitStream(AspAecBatch* b, int stream);
int AspAecBatch_GetControlStream(AspAecBatch* b, int stream, AspAecControl* out);

/* num_frames x (BufferFarend + Process) on [num_frames][num_streams][n] buffers: the loop of
 * test_aec_module.cpp:75-88 in one call (amortises launches; device or host memory). */
int AspAecBatch_Run(AspAecBatch* b, const float* farend, const float* nearend, float* out,
                    int nrOfSamples, int num_frames, int msInSndCardBuf, int mem);
int AspAecBatch_get_echo_status(AspAecBatch* b, int* status /* [num_streams] */);
/* WebRtcAec_GetMetrics for every stream (echo_cancellation.c:456-548), out[num_streams]; metrics
 * are gathered when set_config enabled metricsMode. */
int AspAecBatch_GetMetrics(AspAecBatch* b, AecMetrics* out);
int AspAecBatch_ExportMetricsState(AspAecBatch* b, int stream, AspAecMetricsState* out);
int AspAecBatch_get_error_code(const AspAecBatch* b);
/* WebRtcAec_enable_delay_correction for every stream of the batch: the extended filter (32 partitions,
 * kExtendedMu / kExtendedErrorThreshold, the extended smoothing coefficients and overdrive floors, no filter
 * reset on divergence) and the ProcessExtended / EstBufDelayExtended delay handling of
 * echo_cancellation.c:744-814, 869-922 (trusted-delay build).  Call it after Init (Init switches it off, as
 * aec_core.c:1522-1523 does); the per-stream blocks are re-packed to the new filter length.
 * Restriction: switching it OFF mid-stream is refused (ASP_ERR_STATE, lastError = AEC_UNSUPPORTED_FUNCTION_ERROR,
 * the handle stays in the extended mode) while xfBufBlockPos is 12 or more -- 20 of its 32 positions.  The
 * reference accepts the call there and keeps walking partitions 12..31 of its fixed 32-partition arrays with
 * the 12-partition wrap rule (aec_core.c:1203-1207, 147-169) until the position falls below 12; the 12-partition
 * state block of this library has no such rows.  Feed frames until AspAecBatch_GetControl(...).xfBufBlockPos < 12
 * and call again (a block decrements it by one; the void drop-in wrapper WebRtcAec_enable_delay_correction
 * leaves the code in WebRtcAec_get_error_code). */
int AspAecBatch_enable_delay_correction(AspAecBatch* b, int enable);
int AspAecBatch_delay_correction_enabled(const AspAecBatch* b);
/* WebRtcAec_enable_reported_delay for every stream (aec_core.c:1868-1874).  enable = 0 is the delay-agnostic
 * mode: EstBufDelay is skipped (echo_cancellation.c:725-727, 796-798) and WebRtcAec_ProcessFrames steers every
 * stream's far-end read pointer by that stream's own delay estimate (SignalBasedDelayCorrection,
 * aec_core.c:797-850, 1719-1751) -- the estimator runs when set_config switched delay_logging on.  From the first
 * processed frame on the streams' far buffers move apart; that part of the control plane then runs per stream on the
 * device, and switching reported delays back on needs a new Init. */
int AspAecBatch_enable_reported_delay(AspAecBatch* b, int enable);
int AspAecBatch_reported_delay_enabled(const AspAecBatch* b);
/* WebRtcAec_GetDelayMetrics for every stream (echo_cancellation.c:550-571): median and spread, in ms, of the
 * block-wise delay estimates since the last call (set_config with delay_logging = kAecTrue; -1 / -1 for a
 * stream without estimates; AEC_UNSUPPORTED_FUNCTION_ERROR when logging is off).  median / std [num_streams]. */
int AspAecBatch_GetDelayMetrics(AspAecBatch* b, int* median, int* std);
int AspAecBatch_ExportDelayState(AspAecBatch* b, int stream, AspAecDelayState* out);
int AspAecBatch_ExportState(AspAecBatch* b, int stream, AspAecState* out);
int AspAecBatch_ImportState(AspAecBatch* b, int stream, const AspAecState* in);
int AspAecBatch_GetControl(const AspAecBatch* b, AspAecControl* out);
int AspAecBatch_Synchronize(AspAecBatch* b);
/* Hand-off build of the multi-frame entry points (Run, TimedSteps) with one band and without skew compensation /
 * metrics: the plain configuration; delay logging (the process kernel forms each block's binary far / near spectra and
 * the rest of the estimator runs once per launch over all its blocks); the delay-agnostic mode (each stream's wave
 * runs its own far-buffer control step per sub-frame and the estimator's share of its blocks): the
 * Process launches of up to 64 consecutive frames go
 * into ONE launch (grid y = frame step, its descriptors in device memory), and a per-stream step counter in device
 * memory orders step k + 1 of a stream behind its own step k (every access to the stream's state and far-ring
 * slots write-through / L1-bypassing), so consecutive steps overlap on the chip instead of meeting at a launch
 * boundary.  Same arithmetic, same results bit for bit.  mode: -1 = default (on; ASP_AEC_FLOW=0 in the environment
 * turns the default off), 0 = off, 1 = on.  A wait that times out makes the next synchronising call fail. */
int AspAecBatch_SetFlow(AspAecBatch* b, int mode);
/* `steps` frames of Run over a ring of `frames_in_ring` device frames, bracketed by
 * hipEvents on the launch stream; elapsed_ms of the whole region (bench.py). */
int AspAecBatch_TimedSteps(AspAecBatch* b, const float* farend, const float* nearend, float* out,
                           int nrOfSamples, int frames_in_ring, int steps, float* elapsed_ms);

/* aec_rdft_forward_128 / aec_rdft_inverse_128 (aec_rdft.c:539-556) on `count` rows of 128
 * floats, in place semantics (src -> dst), host pointers.  Parity-test seam. */
int AspAec_rdft128_batch(const float* src, float* dst, int isgn, int count, int device);
/* The binary delay estimator on its own (WebRtc_AddBinaryFarSpectrum + WebRtc_ProcessBinarySpectrum with the robust
 * validation, utility/delay_estimator.c:356-369, 513-644; the logging histogram counts each block's estimate,
 * aec_core.c:1199-1202): `count` independent estimators, each fed its own `nblocks` binary far / near spectra
 * ([count][nblocks], in block order), states in and out through host memory.  Parity-test seam of the kernel the
 * hand-off build runs (utility/delay_estimator_unittest.cc:424-570 restated in tests/test_delay_estimator.py). */
int AspAec_delay_estimator_batch(AspAecDelayState* states, int count, const uint32_t* binary_far, const uint32_t* binary_near,
                                 int nblocks, int device);
/* Host-built constant tables (parity tests compare them with the oracle's and the reference's):
 * which: 0 rdft_w[64], 1 rdft_wk3ri_first[16], 2 rdft_wk3ri_second[16], 3 sqrtHanning[65],
 * 4 weightCurve[65], 5 overDriveCurve[65].  Returns the number of floats written. */
int AspAec_host_table(int which, float* out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* ASP_AEC_H_ */
