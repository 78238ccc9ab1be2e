/*
 * ref_probe_aec.c -- glue compiled INTO oracle/_ref/libaec_ref.so next to the reference's own
 * AEC sources (compiled in place from /root/reference; see oracle/Makefile).  TEST
 * INFRASTRUCTURE ONLY; contains no algorithm.  It forces the reference's plain-C code path
 * (WebRtc_GetCPUInfoNoASM: the SSE2 overrides use a polynomial pow, aec_core_sse2.c:221-355,
 * and are not bit-equal to the C path) and runs the reference entry points the way
 * WebRtc_AMP_Port/test_aec_module.cpp:60-88 does: per 10 ms frame
 *   WebRtcAec_BufferFarend(far, 160); WebRtcAec_Process(near, 1, out, 160, delay_ms, 0).
 */
#include <stdint.h>
#include <string.h>

#include "webrtc/modules/audio_processing/aec/include/echo_cancellation.h"
#include "webrtc/system_wrappers/interface/cpu_features_wrapper.h"

void* ref_aec_create(int32_t fs) {
  void* h = NULL;
  WebRtc_GetCPUInfo = WebRtc_GetCPUInfoNoASM; /* before Create: aec_core.c:1388-1392 */
  if (WebRtcAec_Create(&h) != 0) return NULL;
  if (WebRtcAec_Init(h, fs, 48000) != 0) { /* scSampFreq as in test_aec_module.cpp:61 */
    WebRtcAec_Free(h);
    return NULL;
  }
  return h;
}

void ref_aec_free(void* h) { WebRtcAec_Free(h); }

/* frames: far/near/out [F][160] float (float-S16 range); returns the OR of the
 * return codes of WebRtcAec_Process. */
int ref_aec_run(void* h, const float* far, const float* near, float* out, int F,
                int16_t delay_ms) {
  int rc = 0;
  for (int f = 0; f < F; ++f) {
    float nbuf[160], obuf[160];
    const float* np[1] = {nbuf};
    float* op[1] = {obuf};
    memcpy(nbuf, near + (size_t)f * 160, sizeof nbuf);
    rc |= WebRtcAec_BufferFarend(h, far + (size_t)f * 160, 160);
    rc |= WebRtcAec_Process(h, np, 1, op, 160, delay_ms, 0);
    memcpy(out + (size_t)f * 160, obuf, sizeof obuf);
  }
  return rc;
}

/* ---- state export for the parity tests (field copies only, no algorithm) ---- */
#include "asp_aec.h"
#include "webrtc/common_audio/ring_buffer.h"
#include "webrtc/modules/audio_processing/aec/aec_core_internal.h"
#include "webrtc/modules/audio_processing/aec/echo_cancellation_internal.h"

int ref_aec_frame(void* h, const float* far, const float* near, float* out, int n,
                  int16_t delay_ms) {
  float nbuf[160], obuf[160];
  const float* np[1] = {nbuf};
  float* op[1] = {obuf};
  int rc;
  memcpy(nbuf, near, sizeof(float) * n);
  rc = WebRtcAec_BufferFarend(h, far, (int16_t)n);
  rc |= WebRtcAec_Process(h, np, 1, op, (int16_t)n, delay_ms, 0);
  memcpy(out, obuf, sizeof(float) * n);
  return rc;
}

int ref_aec_set_config(void* h, int mode, int metrics) {
  AecConfig c;
  c.nlpMode = (int16_t)mode;
  c.skewMode = kAecFalse;
  c.metricsMode = (int16_t)metrics;
  c.delay_logging = kAecFalse;
  return WebRtcAec_set_config(h, c);
}

static void copy_level(AspAecPowerLevel* d, const PowerLevel* s) {
  d->sfrsum = s->sfrsum;
  d->sfrcounter = s->sfrcounter;
  d->framelevel = s->framelevel;
  d->frsum = s->frsum;
  d->frcounter = s->frcounter;
  d->minlevel = s->minlevel;
  d->averagelevel = s->averagelevel;
}
static void copy_stats(AspAecStats* d, const Stats* s) {
  d->instant = s->instant;
  d->average = s->average;
  d->min = s->min;
  d->max = s->max;
  d->sum = s->sum;
  d->hisum = s->hisum;
  d->himean = s->himean;
  d->counter = s->counter;
  d->hicounter = s->hicounter;
}
void ref_aec_export_metrics(void* h, AspAecMetricsState* m) {
  const AecCore* k = ((const Aec*)h)->aec;
  copy_level(&m->farlevel, &k->farlevel);
  copy_level(&m->nearlevel, &k->nearlevel);
  copy_level(&m->linoutlevel, &k->linoutlevel);
  copy_level(&m->nlpoutlevel, &k->nlpoutlevel);
  copy_stats(&m->erl, &k->erl);
  copy_stats(&m->erle, &k->erle);
  copy_stats(&m->aNlp, &k->aNlp);
  copy_stats(&m->rerl, &k->rerl);
  m->stateCounter = k->stateCounter;
}
int ref_aec_get_metrics(void* h, AecMetrics* m) { return WebRtcAec_GetMetrics(h, m); }

int ref_aec_set_nlp(void* h, int mode) {
  AecConfig c;
  c.nlpMode = (int16_t)mode;
  c.skewMode = kAecFalse;
  c.metricsMode = kAecFalse;
  c.delay_logging = kAecFalse;
  return WebRtcAec_set_config(h, c);
}

/* the extended filter (32 partitions) is switched on the core, as audio_processing does
 * (WebRtcAec_enable_delay_correction(WebRtcAec_aec_core(handle), 1), aec_core.c:1876-1881) */
void ref_aec_enable_delay_correction(void* h, int enable) {
  WebRtcAec_enable_delay_correction(WebRtcAec_aec_core(h), enable);
}
int ref_aec_delay_correction_enabled(void* h) { return WebRtcAec_delay_correction_enabled(WebRtcAec_aec_core(h)); }

void ref_aec_export(void* h, AspAecState* st, AspAecControl* c) {
  const Aec* a = (const Aec*)h;
  const AecCore* k = a->aec;
  memset(st, 0, sizeof *st);
  memcpy(st->dBuf, k->dBuf, sizeof st->dBuf);
  memcpy(st->eBuf, k->eBuf, sizeof st->eBuf);
  memcpy(st->xPow, k->xPow, sizeof st->xPow);
  memcpy(st->dPow, k->dPow, sizeof st->dPow);
  memcpy(st->dMinPow, k->dMinPow, sizeof st->dMinPow);
  memcpy(st->dInitMinPow, k->dInitMinPow, sizeof st->dInitMinPow);
  memcpy(st->xfBuf[0], k->xfBuf[0], sizeof st->xfBuf[0]);
  memcpy(st->xfBuf[1], k->xfBuf[1], sizeof st->xfBuf[1]);
  memcpy(st->wfBuf[0], k->wfBuf[0], sizeof st->wfBuf[0]);
  memcpy(st->wfBuf[1], k->wfBuf[1], sizeof st->wfBuf[1]);
  memcpy(st->sde, k->sde, sizeof st->sde);
  memcpy(st->sxd, k->sxd, sizeof st->sxd);
  memcpy(st->xfwBuf, k->xfwBuf, sizeof st->xfwBuf);
  memcpy(st->sx, k->sx, sizeof st->sx);
  memcpy(st->sd, k->sd, sizeof st->sd);
  memcpy(st->se, k->se, sizeof st->se);
  memcpy(st->outBuf, k->outBuf, sizeof st->outBuf);
  st->hNlFbMin = k->hNlFbMin;
  st->hNlFbLocalMin = k->hNlFbLocalMin;
  st->hNlXdAvgMin = k->hNlXdAvgMin;
  st->overDrive = k->overDrive;
  st->overDriveSm = k->overDriveSm;
  st->hNlNewMin = k->hNlNewMin;
  st->hNlMinCtr = k->hNlMinCtr;
  st->delayIdx = k->delayIdx;
  st->stNearState = k->stNearState;
  st->echoState = k->echoState;
  st->divergeState = k->divergeState;
  st->xfBufBlockPos = k->xfBufBlockPos;
  st->noiseEstCtr = k->noiseEstCtr;
  st->delayEstCtr = k->delayEstCtr;
  st->seed = k->seed;
  memcpy(st->dBufH, k->dBufH[0], sizeof st->dBufH);
  if (c) {
    memset(c, 0, sizeof *c);
    c->startup_phase = a->startup_phase;
    c->checkBuffSize = a->checkBuffSize;
    c->bufSizeStart = a->bufSizeStart;
    c->knownDelay = a->knownDelay;
    c->filtDelay = a->filtDelay;
    c->timeForDelayChange = a->timeForDelayChange;
    c->lastDelayDiff = a->lastDelayDiff;
    c->counter = a->counter;
    c->sum = a->sum;
    c->firstVal = a->firstVal;
    c->checkBufSizeCtr = a->checkBufSizeCtr;
    c->system_delay = k->system_delay;
    c->core_knownDelay = k->knownDelay;
    /* ring positions are private to ring_buffer.c; the readable counts are the public view */
    c->far_read = (int32_t)WebRtc_available_read(k->far_buf);
    c->pre_read = (int32_t)WebRtc_available_read(a->far_pre_buf);
    c->near_read = (int32_t)WebRtc_available_read(k->nearFrBuf);
    c->out_read = (int32_t)WebRtc_available_read(k->outFrBuf);
  }
}

/* two bands (32 kHz): near / out low and high band of one 10 ms frame */
int ref_aec_frame_bands(void* h, const float* far, const float* near_low, const float* near_high,
                        float* out_low, float* out_high, int n, int16_t delay_ms) {
  float nl[160], nh[160], ol[160], oh[160];
  const float* np[2] = {nl, nh};
  float* op[2] = {ol, oh};
  int rc;
  memcpy(nl, near_low, sizeof(float) * n);
  memcpy(nh, near_high, sizeof(float) * n);
  rc = WebRtcAec_BufferFarend(h, far, (int16_t)n);
  rc |= WebRtcAec_Process(h, np, 2, op, (int16_t)n, delay_ms, 0);
  memcpy(out_low, ol, sizeof(float) * n);
  memcpy(out_high, oh, sizeof(float) * n);
  return rc;
}

void ref_aec_export_dbufh(void* h, float* out128) {
  const Aec* a = (const Aec*)h;
  memcpy(out128, a->aec->dBufH[0], sizeof(float) * 128);
}

/* ---- delay logging, the delay-agnostic mode and skew compensation (field copies / plain calls) ---- */
#include "webrtc/modules/audio_processing/utility/delay_estimator_internal.h"
#include "webrtc/modules/audio_processing/utility/delay_estimator_wrapper.h"

int ref_aec_set_config_full(void* h, int mode, int metrics, int skew_mode, int delay_logging) {
  AecConfig c;
  c.nlpMode = (int16_t)mode;
  c.skewMode = (int16_t)skew_mode;
  c.metricsMode = (int16_t)metrics;
  c.delay_logging = (int16_t)delay_logging;
  return WebRtcAec_set_config(h, c);
}
void ref_aec_enable_reported_delay(void* h, int enable) {
  WebRtcAec_enable_reported_delay(WebRtcAec_aec_core(h), enable);
}
int ref_aec_reported_delay_enabled(void* h) { return WebRtcAec_reported_delay_enabled(WebRtcAec_aec_core(h)); }
int ref_aec_get_delay_metrics(void* h, int* median, int* std) { return WebRtcAec_GetDelayMetrics(h, median, std); }
int ref_aec_error_code(void* h) { return WebRtcAec_get_error_code(h); }

/* one frame with a skew argument (WebRtcAec_Process's last parameter) */
int ref_aec_frame_skew(void* h, const float* far, const float* near, float* out, int n, int16_t delay_ms,
                       int32_t skew) {
  float nbuf[160], obuf[160];
  const float* np[1] = {nbuf};
  float* op[1] = {obuf};
  int rc;
  memcpy(nbuf, near, sizeof(float) * n);
  rc = WebRtcAec_BufferFarend(h, far, (int16_t)n);
  rc |= WebRtcAec_Process(h, np, 1, op, (int16_t)n, delay_ms, skew);
  memcpy(out, obuf, sizeof(float) * n);
  return rc;
}
void ref_aec_export_skew(void* h, float* skew, int* resample, int* skewFrCtr) {
  const Aec* a = (const Aec*)h;
  *skew = a->skew;
  *resample = a->resample;
  *skewFrCtr = a->skewFrCtr;
}

/* far_read carries the readable count of far_buf (ring positions are private to ring_buffer.c);
 * far_write / far_wrap are -1 */
void ref_aec_export_delay(void* h, AspAecDelayState* d) {
  const AecCore* k = ((const Aec*)h)->aec;
  const DelayEstimatorFarend* fe = (const DelayEstimatorFarend*)k->delay_estimator_farend;
  const DelayEstimator* ne = (const DelayEstimator*)k->delay_estimator;
  const BinaryDelayEstimatorFarend* bf = fe->binary_farend;
  const BinaryDelayEstimator* bn = ne->binary_handle;
  int i;
  memset(d, 0, sizeof *d);
  for (i = 0; i < 65; ++i) d->mean_far_spectrum[i] = fe->mean_far_spectrum[i].float_;
  d->far_spectrum_initialized = fe->far_spectrum_initialized;
  memcpy(d->binary_far_history, bf->binary_far_history, sizeof d->binary_far_history);
  for (i = 0; i < ASP_AEC_DELAY_HISTORY; ++i) d->far_bit_counts[i] = bf->far_bit_counts[i];
  for (i = 0; i < 65; ++i) d->mean_near_spectrum[i] = ne->mean_near_spectrum[i].float_;
  d->near_spectrum_initialized = ne->near_spectrum_initialized;
  memcpy(d->binary_near_history, bn->binary_near_history, sizeof d->binary_near_history);
  memcpy(d->mean_bit_counts, bn->mean_bit_counts, sizeof d->mean_bit_counts);
  memcpy(d->bit_counts, bn->bit_counts, sizeof d->bit_counts);
  memcpy(d->histogram, bn->histogram, sizeof d->histogram);
  d->minimum_probability = bn->minimum_probability;
  d->last_delay_probability = bn->last_delay_probability;
  d->last_delay = bn->last_delay;
  d->last_candidate_delay = bn->last_candidate_delay;
  d->compare_delay = bn->compare_delay;
  d->candidate_hits = bn->candidate_hits;
  d->last_delay_histogram = bn->last_delay_histogram;
  d->lookahead = bn->lookahead;
  d->allowed_offset = bn->allowed_offset;
  memcpy(d->delay_histogram, k->delay_histogram, sizeof d->delay_histogram);
  d->previous_delay = k->previous_delay;
  d->delay_correction_count = k->delay_correction_count;
  d->shift_offset = k->shift_offset;
  d->delay_quality_threshold = k->delay_quality_threshold;
  d->far_read = (int32_t)WebRtc_available_read(k->far_buf);
  d->far_write = -1;
  d->far_wrap = -1;
  d->system_delay = k->system_delay;
}

/* the two entry points called separately (the reference's system_delay_unittest.cc drives them so) */
int ref_aec_buffer_farend(void* h, const float* far, int n) { return WebRtcAec_BufferFarend(h, far, (int16_t)n); }
int ref_aec_process(void* h, const float* near, float* out, int n, int16_t delay_ms) {
  float nbuf[160], obuf[160];
  const float* np[1] = {nbuf};
  float* op[1] = {obuf};
  int rc;
  memcpy(nbuf, near, sizeof(float) * n);
  rc = WebRtcAec_Process(h, np, 1, op, (int16_t)n, delay_ms, 0);
  memcpy(out, obuf, sizeof(float) * n);
  return rc;
}
