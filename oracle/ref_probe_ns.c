/*
 * ref_probe_ns.c -- glue compiled INTO oracle/_ref/libns_ref.so next to the
 * reference's own ns_core.c / noise_suppression.c / fft4g.c (which are compiled
 * in place from /root/reference; see oracle/Makefile).  TEST INFRASTRUCTURE
 * ONLY.  It contains no algorithm: it converts the reference's
 * NoiseSuppressionC (ns/ns_core.h:52-114) to and from the canonical
 * AspNsState of include/asp_ns.h so the tests can snapshot / inject reference
 * state, and runs the reference's own entry points over a batch of streams
 * for the golden-vector generator and the CPU baseline.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "asp_ns.h"
#include "webrtc/modules/audio_processing/ns/ns_core.h"
#include "webrtc/modules/audio_processing/utility/fft4g.h"

#define CP(dst, src) memcpy((dst), (src), sizeof(dst))

size_t ref_ns_sizeof(void) { return sizeof(NoiseSuppressionC); }

void ref_ns_export(const NoiseSuppressionC* r, AspNsState* s) {
  memset(s, 0, sizeof *s);
  s->fs = (int32_t)r->fs;
  s->aggrMode = r->aggrMode;
  s->initFlag = r->initFlag;
  s->gainmap = r->gainmap;
  s->blockInd = r->blockInd;
  s->updates = r->updates;
  CP(s->counter, r->counter);
  CP(s->modelUpdatePars, r->modelUpdatePars);
  s->overdrive = r->overdrive;
  s->denoiseBound = r->denoiseBound;
  s->priorSpeechProb = r->priorSpeechProb;
  s->signalEnergy = r->signalEnergy;
  s->sumMagn = r->sumMagn;
  s->whiteNoiseLevel = r->whiteNoiseLevel;
  s->pinkNoiseNumerator = r->pinkNoiseNumerator;
  s->pinkNoiseExp = r->pinkNoiseExp;
  CP(s->priorModelPars, r->priorModelPars);
  CP(s->featureData, r->featureData);
  CP(s->analyzeBuf, r->analyzeBuf);
  CP(s->dataBuf, r->dataBuf);
  CP(s->syntBuf, r->syntBuf);
  CP(s->density, r->density);
  CP(s->lquantile, r->lquantile);
  CP(s->quantile, r->quantile);
  CP(s->smooth, r->smooth);
  CP(s->noise, r->noise);
  CP(s->noisePrev, r->noisePrev);
  CP(s->magnPrevAnalyze, r->magnPrevAnalyze);
  CP(s->magnPrevProcess, r->magnPrevProcess);
  CP(s->logLrtTimeAvg, r->logLrtTimeAvg);
  CP(s->magnAvgPause, r->magnAvgPause);
  CP(s->initMagnEst, r->initMagnEst);
  CP(s->parametricNoise, r->parametricNoise);
  CP(s->speechProb, r->speechProb);
  CP(s->histLrt, r->histLrt);
  CP(s->histSpecFlat, r->histSpecFlat);
  CP(s->histSpecDiff, r->histSpecDiff);
}

/* `r` must already have been through WebRtcNs_InitCore (window pointer, FFT
 * work arrays, feature-extraction constants are left as Init set them). */
void ref_ns_import(NoiseSuppressionC* r, const AspNsState* s) {
  r->aggrMode = s->aggrMode;
  r->gainmap = s->gainmap;
  r->blockInd = s->blockInd;
  r->updates = s->updates;
  CP(r->counter, s->counter);
  CP(r->modelUpdatePars, s->modelUpdatePars);
  r->overdrive = s->overdrive;
  r->denoiseBound = s->denoiseBound;
  r->priorSpeechProb = s->priorSpeechProb;
  r->signalEnergy = s->signalEnergy;
  r->sumMagn = s->sumMagn;
  r->whiteNoiseLevel = s->whiteNoiseLevel;
  r->pinkNoiseNumerator = s->pinkNoiseNumerator;
  r->pinkNoiseExp = s->pinkNoiseExp;
  CP(r->priorModelPars, s->priorModelPars);
  CP(r->featureData, s->featureData);
  CP(r->analyzeBuf, s->analyzeBuf);
  CP(r->dataBuf, s->dataBuf);
  CP(r->syntBuf, s->syntBuf);
  CP(r->density, s->density);
  CP(r->lquantile, s->lquantile);
  CP(r->quantile, s->quantile);
  CP(r->smooth, s->smooth);
  CP(r->noise, s->noise);
  CP(r->noisePrev, s->noisePrev);
  CP(r->magnPrevAnalyze, s->magnPrevAnalyze);
  CP(r->magnPrevProcess, s->magnPrevProcess);
  CP(r->logLrtTimeAvg, s->logLrtTimeAvg);
  CP(r->magnAvgPause, s->magnAvgPause);
  CP(r->initMagnEst, s->initMagnEst);
  CP(r->parametricNoise, s->parametricNoise);
  CP(r->speechProb, s->speechProb);
  CP(r->histLrt, s->histLrt);
  CP(r->histSpecFlat, s->histSpecFlat);
  CP(r->histSpecDiff, s->histSpecDiff);
}

/* FFT work-array view, to pin the twiddle tables. */
void ref_ns_fft_tables(const NoiseSuppressionC* r, int* ip, float* w) {
  memcpy(ip, r->ip, sizeof r->ip);
  memcpy(w, r->wfft, sizeof r->wfft);
}

/* WebRtc_rdft on one 256-float row with freshly initialised work arrays. */
void ref_rdft256(float* a, int isgn) {
  static __thread int ip[128];
  static __thread float w[128];
  static __thread int ready;
  if (!ready) {
    float z[256];
    memset(z, 0, sizeof z);
    ip[0] = 0;
    WebRtc_rdft(256, 1, z, ip, w);
    ready = 1;
  }
  WebRtc_rdft(256, isgn, a, ip, w);
}

/* WebRtc_rdft on one 128-float row (the 8 kHz transform) with freshly initialised work arrays. */
void ref_rdft128(float* a, int isgn) {
  static __thread int ip[128];
  static __thread float w[128];
  static __thread int ready;
  if (!ready) {
    float z[128];
    memset(z, 0, sizeof z);
    ip[0] = 0;
    WebRtc_rdft(128, 1, z, ip, w);
    ready = 1;
  }
  WebRtc_rdft(128, isgn, a, ip, w);
}

/* Batch loop of the reference entry points: frames [F][S][160], Analyze then
 * Process on the same frame per stream (test_ns_module.cpp:97-99). */
typedef struct {
  NoiseSuppressionC* inst;
  int s0, s1, S, F;
  const float* in;
  float* out;
} RefShard;

static void* ref_shard_main(void* p) {
  RefShard* sh = (RefShard*)p;
  const size_t bl = (size_t)sh->inst[0].blockLen; /* 160, or 80 at 8 kHz: frames are [F][S][blockLen] */
  for (int f = 0; f < sh->F; ++f)
    for (int st = sh->s0; st < sh->s1; ++st) {
      float tmp[160], o[160];
      const float* ip[1] = {tmp};
      float* op[1] = {o};
      memcpy(tmp, sh->in + ((size_t)f * sh->S + st) * bl, sizeof(float) * bl);
      WebRtcNs_AnalyzeCore(&sh->inst[st], tmp);
      WebRtcNs_ProcessCore(&sh->inst[st], ip, 1, op);
      memcpy(sh->out + ((size_t)f * sh->S + st) * bl, o, sizeof(float) * bl);
    }
  return NULL;
}

void ref_ns_run(NoiseSuppressionC* inst, int S, const float* in, float* out,
                int F, int threads) {
  if (threads < 1) threads = 1;
  if (threads > S) threads = S;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  RefShard* sh = (RefShard*)malloc(sizeof(RefShard) * (size_t)threads);
  for (int t = 0; t < threads; ++t) {
    sh[t].inst = inst;
    sh[t].s0 = (int)((long long)S * t / threads);
    sh[t].s1 = (int)((long long)S * (t + 1) / threads);
    sh[t].S = S;
    sh[t].F = F;
    sh[t].in = in;
    sh[t].out = out;
    pthread_create(&th[t], NULL, ref_shard_main, &sh[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th);
  free(sh);
}

/* ---- multi-band runs (32 / 48 kHz: one or two high bands next to the low band) ---- */
/* One stream: frames low [F][160], high [F][num_high][160]; Analyze(low) then
 * Process(bands, 1 + num_high) per frame, the order of libapm/src/apm_ns.cpp:69-74. */
void ref_ns_run_bands(NoiseSuppressionC* inst, const float* low, const float* high, int num_high,
                      float* out_low, float* out_high, int F) {
  for (int f = 0; f < F; ++f) {
    float lb[160], hb[2][160], ol[160], oh[2][160];
    const float* in_b[3] = {lb, hb[0], hb[1]};
    float* out_b[3] = {ol, oh[0], oh[1]};
    memcpy(lb, low + (size_t)f * 160, sizeof lb);
    for (int i = 0; i < num_high; ++i)
      memcpy(hb[i], high + ((size_t)f * num_high + i) * 160, sizeof hb[i]);
    WebRtcNs_AnalyzeCore(inst, lb);
    WebRtcNs_ProcessCore(inst, in_b, 1 + num_high, out_b);
    memcpy(out_low + (size_t)f * 160, ol, sizeof ol);
    for (int i = 0; i < num_high; ++i)
      memcpy(out_high + ((size_t)f * num_high + i) * 160, oh[i], sizeof oh[i]);
  }
}

void ref_ns_export_hb(const NoiseSuppressionC* inst, AspNsHbState* out) {
  memcpy(out->dataBufHB, inst->dataBufHB, sizeof out->dataBufHB);
}
