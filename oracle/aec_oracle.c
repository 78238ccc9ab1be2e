/*
 * aec_oracle.c -- CPU restatement of the reference's WebRTC AEC.
 * TEST INFRASTRUCTURE ONLY (see aec_oracle.h).  Parity: PINNED against the reference build
 * oracle/_ref/libaec_ref.so (bit-exact, tests/test_aec_oracle.py).
 *
 * Paths below are relative to WebRtc_AMP_Port/webrtc/ ; "core" = modules/audio_processing/aec/
 * aec_core.c, "ec" = .../aec/echo_cancellation.c, "rdft" = .../aec/aec_rdft.c,
 * "ring" = common_audio/ring_buffer.c, "de" / "dw" = modules/audio_processing/utility/delay_estimator.c /
 * delay_estimator_wrapper.c, "rs" = .../aec/aec_resampler.c.
 *
 * Covered configuration: what test_aec_module.cpp:60-88 runs (one band at 8 / 16 kHz, 12 partitions,
 * reported-delay mode) plus the second band at 32 kHz, echo metrics, the extended filter, delay logging with
 * WebRtcAec_GetDelayMetrics, the delay-agnostic mode (reported delays off) and skew compensation -- each pinned
 * to the reference build by its own test in tests/test_aec_oracle.py.
 * Compile with -ffp-contract=off.
 */
#include "aec_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PART_LEN 64
#define PART_LEN1 65
#define PART_LEN2 128
#define FRAME_LEN 80
#define NPART_NORMAL 12   /* kNormalNumPartitions, core_internal:25 */
#define NPART_MAX 32      /* kExtendedNumPartitions, core_internal:23: the arrays are always this long */
#define FAR_SLOTS 250
#define PRE_LEN (PART_LEN2 + 4 * FRAME_LEN) /* ec:146-147, aec_resampler.h:20 */
#define FRBUF_LEN (FRAME_LEN + PART_LEN)     /* core:1299 */

/* ------------------------------------------------------------------ tables */
static float g_w[64];        /* rdft_w,            rdft:32-49  */
static float g_wk3_a[16];    /* rdft_wk3ri_first,  rdft:50-55  */
static float g_wk3_b[16];    /* rdft_wk3ri_second, rdft:56-61  */
static float g_hann[65];     /* WebRtcAec_sqrtHanning, core:53-70   */
static float g_weight[65];   /* WebRtcAec_weightCurve, core:75-84   */
static float g_odrive[65];   /* WebRtcAec_overDriveCurve, core:89-98 */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static unsigned bitrev(unsigned x, int bits) {
  unsigned r = 0;
  for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1u) << (bits - 1 - b);
  return r;
}

/* A table printed by Matlab's fprintf('%.<d>f') and re-read by the C compiler as a float
 * literal: reproduce the decimal round trip. */
static float via_text(double v, int decimals) {
  char buf[48];
  snprintf(buf, sizeof buf, "%.*f", decimals, v);
  return strtof(buf, NULL);
}

static void build_tables(void) {
  /* The reference's tables "used to be computed at run-time" (rdft:29-31) by Ooura's makewt /
   * makect for n = 128 (utility/fft4g.c:642-690 is the same code at n = 256): */
  float tmp[32];
  const int nw = 32, nwh = 16, nc = 32, nch = 16;
  float delta = (float)atan(1.0f) / nwh;
  tmp[0] = 1;
  tmp[1] = 0;
  tmp[nwh] = (float)cos(delta * nwh);
  tmp[nwh + 1] = tmp[nwh];
  for (int j = 2; j < nwh; j += 2) {
    float x = (float)cos(delta * j);
    float y = (float)sin(delta * j);
    tmp[j] = x;
    tmp[j + 1] = y;
    tmp[nw - j] = y;
    tmp[nw - j + 1] = x;
  }
  for (int j = 0; j < 16; ++j) { /* bitrv2(nw, ...) = bit reversal of the 16 complex entries */
    unsigned r = bitrev((unsigned)j, 4);
    g_w[2 * j] = tmp[2 * r];
    g_w[2 * j + 1] = tmp[2 * r + 1];
  }
  delta = (float)atan(1.0f) / nch;
  g_w[32] = (float)cos(delta * nch);
  g_w[32 + nch] = 0.5f * g_w[32];
  for (int j = 1; j < nch; j++) {
    g_w[32 + j] = 0.5f * (float)cos(delta * j);
    g_w[32 + nc - j] = 0.5f * (float)sin(delta * j);
  }
  /* The reference freezes the table as decimal text (rdft:32-49); against makewt / makect
   * evaluated with this libm the text differs by one unit in the last place at eight entries.
   * That is data of the reference, applied here as data: */
  {
    static const signed char nudge[8][2] = {{4, 1}, {7, 1}, {20, 1}, {27, 1},
                                            {40, 1}, {41, -1}, {42, 1}, {47, 1}};
    for (int k = 0; k < 8; ++k) {
      int32_t bits;
      memcpy(&bits, &g_w[nudge[k][0]], sizeof bits);
      bits += nudge[k][1];
      memcpy(&g_w[nudge[k][0]], &bits, sizeof bits);
    }
  }
  /* wk3 = wk1 - 2 wk2i wk1i etc. (fft4g.c:1054-1055,1081-1082), tabulated by the reference */
  for (int k1 = 0; k1 < 16; k1 += 2) {
    const int k2 = 2 * k1;
    const float wk2r = g_w[k1], wk2i = g_w[k1 + 1];
    float wk1r = g_w[k2], wk1i = g_w[k2 + 1];
    g_wk3_a[k1] = wk1r - 2 * wk2i * wk1i;
    g_wk3_a[k1 + 1] = 2 * wk2i * wk1r - wk1i;
    wk1r = g_w[k2 + 2];
    wk1i = g_w[k2 + 3];
    g_wk3_b[k1] = wk1r - 2 * wk2r * wk1i;
    g_wk3_b[k1 + 1] = 2 * wk2r * wk1r - wk1i;
  }
  /* core:50-70: square root of a Hann window, first half: 65 entries sin(pi k / 128) printed
   * %.14f */
  for (int k = 0; k <= 64; ++k) g_hann[k] = via_text(sin(M_PI * k / 128.0), 14);
  /* core:72-74: weightCurve = [0 ; 0.3 * sqrt(linspace(0,1,64))' + 0.1], printed %.4f */
  g_weight[0] = 0.f;
  for (int k = 1; k <= 64; ++k) g_weight[k] = via_text(0.3 * sqrt((k - 1) / 63.0) + 0.1, 4);
  /* core:86-88: overDriveCurve = sqrt(linspace(0,1,65))' + 1, printed %.4f */
  for (int k = 0; k <= 64; ++k) g_odrive[k] = via_text(sqrt(k / 64.0) + 1.0, 4);
}

static void ensure_tables(void) { pthread_once(&g_once, build_tables); }

const float* asp_aec_oracle_table(int which) {
  ensure_tables();
  switch (which) {
    case 0: return g_w;
    case 1: return g_wk3_a;
    case 2: return g_wk3_b;
    case 3: return g_hann;
    case 4: return g_weight;
    case 5: return g_odrive;
    default: return NULL;
  }
}

/* --------------------------------------------------------------------- FFT */
/* One radix-4 pass over 64 complex points at complex stride l: cft1st_128 (l = 1, rdft:201-311)
 * and cftmdl_128 (l = 4, rdft:313-444).  Blocks of 4l points, B = 0: no twiddles; B = 1: the
 * w[2] block; B = 2u / 2u+1: twiddles w[2u], w[4u] (+2) and the tabulated wk3. */
static void cft_pass64(float* a, int l) {
  const int N = 64, bs = 4 * l;
  for (int B = 0; B * bs < N; ++B) {
    int kind = 2;
    float ws = 0.f, w1r = 0.f, w1i = 0.f, w2r = 0.f, w2i = 0.f, w3r = 0.f, w3i = 0.f;
    if (B == 0) {
      kind = 0;
    } else if (B == 1) {
      kind = 1;
      ws = g_w[2];
    } else {
      const int k1 = 2 * (B >> 1), k2 = 2 * k1;
      const float wk2r = g_w[k1], wk2i = g_w[k1 + 1];
      if ((B & 1) == 0) {
        w1r = g_w[k2];
        w1i = g_w[k2 + 1];
        w3r = g_wk3_a[k1];
        w3i = g_wk3_a[k1 + 1];
        w2r = wk2r;
        w2i = wk2i;
      } else {
        w1r = g_w[k2 + 2];
        w1i = g_w[k2 + 3];
        w3r = g_wk3_b[k1];
        w3i = g_wk3_b[k1 + 1];
        w2r = -wk2i;
        w2i = wk2r;
      }
    }
    for (int q = 0; q < l; ++q) {
      float* e0 = a + 2 * (B * bs + q);
      float* e1 = e0 + 2 * l;
      float* e2 = e1 + 2 * l;
      float* e3 = e2 + 2 * l;
      float x0r = e0[0] + e1[0], x0i = e0[1] + e1[1];
      float x1r = e0[0] - e1[0], x1i = e0[1] - e1[1];
      float x2r = e2[0] + e3[0], x2i = e2[1] + e3[1];
      float x3r = e2[0] - e3[0], x3i = e2[1] - e3[1];
      e0[0] = x0r + x2r;
      e0[1] = x0i + x2i;
      if (kind == 0) {
        e2[0] = x0r - x2r;
        e2[1] = x0i - x2i;
        e1[0] = x1r - x3i;
        e1[1] = x1i + x3r;
        e3[0] = x1r + x3i;
        e3[1] = x1i - x3r;
      } else if (kind == 1) {
        float yr, yi;
        e2[0] = x2i - x0i;
        e2[1] = x0r - x2r;
        yr = x1r - x3i;
        yi = x1i + x3r;
        e1[0] = ws * (yr - yi);
        e1[1] = ws * (yr + yi);
        yr = x3i + x1r;
        yi = x3r - x1i;
        e3[0] = ws * (yi - yr);
        e3[1] = ws * (yi + yr);
      } else {
        float yr, yi;
        x0r -= x2r;
        x0i -= x2i;
        e2[0] = w2r * x0r - w2i * x0i;
        e2[1] = w2r * x0i + w2i * x0r;
        yr = x1r - x3i;
        yi = x1i + x3r;
        e1[0] = w1r * yr - w1i * yi;
        e1[1] = w1r * yi + w1i * yr;
        yr = x1r + x3i;
        yi = x1i - x3r;
        e3[0] = w3r * yr - w3i * yi;
        e3[1] = w3r * yi + w3i * yr;
      }
    }
  }
}

/* bitrv2_128 (rdft:124-199) is the bit reversal of the 64 complex indices. */
static void bit_reverse64(float* a) {
  for (int i = 0; i < 64; ++i) {
    const int r = (int)bitrev((unsigned)i, 6);
    if (r > i) {
      const float tr = a[2 * i], ti = a[2 * i + 1];
      a[2 * i] = a[2 * r];
      a[2 * i + 1] = a[2 * r + 1];
      a[2 * r] = tr;
      a[2 * r + 1] = ti;
    }
  }
}

/* cftfsub_128 / cftbsub_128 (rdft:446-507): two radix-4 passes and the twiddle-free last one. */
static void cft64(float* a, int backward) {
  cft_pass64(a, 1);
  cft_pass64(a, 4);
  for (int q = 0; q < 16; ++q) {
    float* e0 = a + 2 * q;
    float* e1 = e0 + 32;
    float* e2 = e1 + 32;
    float* e3 = e2 + 32;
    if (!backward) {
      const float x0r = e0[0] + e1[0], x0i = e0[1] + e1[1];
      const float x1r = e0[0] - e1[0], x1i = e0[1] - e1[1];
      const float x2r = e2[0] + e3[0], x2i = e2[1] + e3[1];
      const float x3r = e2[0] - e3[0], x3i = e2[1] - e3[1];
      e0[0] = x0r + x2r;
      e0[1] = x0i + x2i;
      e2[0] = x0r - x2r;
      e2[1] = x0i - x2i;
      e1[0] = x1r - x3i;
      e1[1] = x1i + x3r;
      e3[0] = x1r + x3i;
      e3[1] = x1i - x3r;
    } else {
      const float x0r = e0[0] + e1[0], x0i = -e0[1] - e1[1];
      const float x1r = e0[0] - e1[0], x1i = -e0[1] + e1[1];
      const float x2r = e2[0] + e3[0], x2i = e2[1] + e3[1];
      const float x3r = e2[0] - e3[0], x3i = e2[1] - e3[1];
      e0[0] = x0r + x2r;
      e0[1] = x0i - x2i;
      e2[0] = x0r - x2r;
      e2[1] = x0i + x2i;
      e1[0] = x1r - x3i;
      e1[1] = x1i - x3r;
      e3[0] = x1r + x3i;
      e3[1] = x1i + x3r;
    }
  }
}

void asp_aec_oracle_rdft128(float* a, int isgn) {
  const float* c = g_w + 32;
  ensure_tables();
  if (isgn >= 0) { /* aec_rdft_forward_128, rdft:539-547 */
    float xi;
    bit_reverse64(a);
    cft64(a, 0);
    for (int j1 = 1, j2 = 2; j2 < 64; j1 += 1, j2 += 2) { /* rftfsub_128, rdft:509-527 */
      const int k2 = 128 - j2, k1 = 32 - j1;
      const float wkr = 0.5f - c[k1], wki = c[j1];
      const float xr = a[j2] - a[k2], xim = a[j2 + 1] + a[k2 + 1];
      const float yr = wkr * xr - wki * xim, yi = wkr * xim + wki * xr;
      a[j2] -= yr;
      a[j2 + 1] -= yi;
      a[k2] += yr;
      a[k2 + 1] -= yi;
    }
    xi = a[0] - a[1];
    a[0] += a[1];
    a[1] = xi;
  } else { /* aec_rdft_inverse_128, rdft:549-556 */
    a[1] = 0.5f * (a[0] - a[1]);
    a[0] -= a[1];
    a[1] = -a[1]; /* rftbsub_128, rdft:529-537 */
    for (int j1 = 1, j2 = 2; j2 < 64; j1 += 1, j2 += 2) {
      const int k2 = 128 - j2, k1 = 32 - j1;
      const float wkr = 0.5f - c[k1], wki = c[j1];
      const float xr = a[j2] - a[k2], xim = a[j2 + 1] + a[k2 + 1];
      const float yr = wkr * xr + wki * xim, yi = wkr * xim - wki * xr;
      a[j2] = a[j2] - yr;
      a[j2 + 1] = yi - a[j2 + 1];
      a[k2] = yr + a[k2];
      a[k2 + 1] = yi - a[k2 + 1];
    }
    a[65] = -a[65];
    bit_reverse64(a);
    cft64(a, 1);
  }
}

/* -------------------------------------------------------------- ring buffers */
/* ring:20-249 with positions kept apart from the data (the batched build keeps exactly this
 * integer part on the host). */
typedef struct RingPos {
  int read, write, wrap, count; /* wrap: 0 SAME_WRAP, 1 DIFF_WRAP */
} RingPos;

static void rp_init(RingPos* r, int count) {
  r->read = 0;
  r->write = 0;
  r->wrap = 0;
  r->count = count;
}
static int rp_avail_read(const RingPos* r) { /* ring:231-240 */
  return r->wrap == 0 ? r->write - r->read : r->count - r->read + r->write;
}
static int rp_avail_write(const RingPos* r) { return r->count - rp_avail_read(r); }
static int rp_move_read(RingPos* r, int n) { /* ring:195-228 */
  const int free_elements = rp_avail_write(r), readable = rp_avail_read(r);
  int pos = r->read;
  if (n > readable) n = readable;
  if (n < -free_elements) n = -free_elements;
  pos += n;
  if (pos > r->count) {
    pos -= r->count;
    r->wrap = 0;
  }
  if (pos < 0) {
    pos += r->count;
    r->wrap = 1;
  }
  r->read = pos;
  return n;
}
static int ring_write(RingPos* r, float* data, int ef, const float* src, int n) { /* ring:161-192 */
  const int free_elements = rp_avail_write(r);
  const int write_elements = free_elements < n ? free_elements : n;
  int m = write_elements;
  const int margin = r->count - r->write;
  if (write_elements > margin) {
    memcpy(data + (size_t)r->write * ef, src, (size_t)margin * ef * sizeof(float));
    r->write = 0;
    m -= margin;
    r->wrap = 1;
  }
  memcpy(data + (size_t)r->write * ef, src + (size_t)(write_elements - m) * ef,
         (size_t)m * ef * sizeof(float));
  r->write += m;
  return write_elements;
}
static int ring_read(RingPos* r, const float* data, int ef, float* dst, int n) { /* ring:37-60,112-158 */
  const int readable = rp_avail_read(r);
  const int read_elements = readable < n ? readable : n;
  const int margin = r->count - r->read;
  if (read_elements > margin) {
    memcpy(dst, data + (size_t)r->read * ef, (size_t)margin * ef * sizeof(float));
    memcpy(dst + (size_t)margin * ef, data, (size_t)(read_elements - margin) * ef * sizeof(float));
  } else {
    memcpy(dst, data + (size_t)r->read * ef, (size_t)read_elements * ef * sizeof(float));
  }
  rp_move_read(r, read_elements);
  return read_elements;
}

/* ------------------------------------------------------------------ object */
enum { RS_BUF = 4 * FRAME_LEN, RS_EST = 400, RS_DELAY = 1 }; /* kResamplerBufferSize, kEstimateLengthFrames, kResamplingDelay */
struct AspAecOracle {
  AspAecState st;
  /* Aec (ec, echo_cancellation_internal.h:17-65) */
  int sampFreq, scSampFreq, splitSampFreq, rate_factor, initFlag, lastError, farend_started;
  int skewMode;
  int bufSizeStart, knownDelay, timeForDelayChange, startup_phase, checkBuffSize, sum;
  int16_t counter, firstVal, checkBufSizeCtr, msInSndCardBuf, filtDelay, lastDelayDiff;
  /* integer part of AecCore */
  int system_delay, core_knownDelay, mult, nlp_mode, metricsMode, delay_logging;
  float normal_mu, normal_error_threshold;
  int extended_filter_enabled, num_partitions; /* WebRtcAec_enable_delay_correction, core:1876-1881 */
  int reported_delay_enabled;                  /* WebRtcAec_enable_reported_delay, core:1868-1874 */
  AspAecDelayState de;                         /* delay estimator + AecCore's fields around it */
  /* skew compensation (ec:304-315, 606-645; aec_resampler.c) */
  int16_t skewFrCtr;
  int resample, highSkewCtr;
  float skew, sampFactor;
  float rs_buffer[RS_BUF];
  float rs_position;
  int rs_deviceSampleRateHz;
  int rs_skewData[RS_EST];
  int rs_skewDataIndex;
  float rs_skewEstimate;
  int blocks_processed;
  RingPos pre_pos, far_pos, near_pos, out_pos;
  float pre[PRE_LEN];
  float far[FAR_SLOTS][2 * PART_LEN1];
  float farw[FAR_SLOTS][2 * PART_LEN1];
  float nearfr[FRBUF_LEN];
  float outfr[FRBUF_LEN];
  AspAecMetricsState met; /* metricsMode = 1: core:548-770 */
  /* 32 kHz: one high band next to the low band (its rings move in lock-step with the low band's) */
  int num_bands;
  float nearfrH[FRBUF_LEN];
  float outfrH[FRBUF_LEN];
};

static const int kInitCheck = 42;      /* ec:59 */
static const int kMaxTrustedDelayMs = 500; /* ec:53 */
static const int kMaxBufSizeStart = 62;    /* ec:57 */
static const int kSampMsNb = 8;            /* ec:58 */

AspAecOracle* asp_aec_oracle_create(void) { /* ec:121-168 */
  AspAecOracle* o = (AspAecOracle*)calloc(1, sizeof *o);
  ensure_tables();
  if (o) {
    o->initFlag = 0;
    o->lastError = 0;
    o->de.lookahead = ASP_AEC_DELAY_LOOKAHEAD; /* WebRtc_set_lookahead at Create, core:1374-1378 (not Android) */
  }
  return o;
}

void asp_aec_oracle_free(AspAecOracle* o) { free(o); }

static int far_move_read(AspAecOracle* o, int elements) { /* WebRtcAec_MoveFarReadPtr, core:1637-1645 */
  const int moved = rp_move_read(&o->far_pos, elements);
  o->system_delay -= moved * PART_LEN;
  return moved;
}

static void init_level(AspAecPowerLevel* l) { /* InitLevel, core:548-558 */
  l->averagelevel = 0;
  l->framelevel = 0;
  l->minlevel = 1E17f;
  l->frsum = 0;
  l->sfrsum = 0;
  l->frcounter = 0;
  l->sfrcounter = 0;
}
static void init_stats(AspAecStats* s) { /* InitStats, core:560-570 */
  s->instant = -100;
  s->average = -100;
  s->max = -100;
  s->min = 100;
  s->sum = 0;
  s->hisum = 0;
  s->himean = -100;
  s->counter = 0;
  s->hicounter = 0;
}
static void init_metrics(AspAecMetricsState* m) { /* InitMetrics, core:572-583 */
  m->stateCounter = 0;
  init_level(&m->farlevel);
  init_level(&m->nearlevel);
  init_level(&m->linoutlevel);
  init_level(&m->nlpoutlevel);
  init_stats(&m->erl);
  init_stats(&m->erle);
  init_stats(&m->aNlp);
  init_stats(&m->rerl);
}

static void update_level(AspAecPowerLevel* level, float in[2][PART_LEN1]) { /* core:585-642 */
  const int subCountLen = 4, countLen = 50;
  float energy = (in[0][0] * in[0][0]) / 2;
  energy += (in[0][PART_LEN] * in[0][PART_LEN]) / 2;
  for (int k = 1; k < PART_LEN; k++) energy += (in[0][k] * in[0][k] + in[1][k] * in[1][k]);
  energy /= PART_LEN2;
  level->sfrsum += energy;
  level->sfrcounter++;
  if (level->sfrcounter > subCountLen) {
    level->framelevel = level->sfrsum / (subCountLen * PART_LEN);
    level->sfrsum = 0;
    level->sfrcounter = 0;
    if (level->framelevel > 0) {
      if (level->framelevel < level->minlevel) {
        level->minlevel = level->framelevel;
      } else {
        level->minlevel *= (1 + 0.001f);
      }
    }
    level->frcounter++;
    level->frsum += level->framelevel;
    if (level->frcounter > countLen) {
      level->averagelevel = level->frsum / countLen;
      level->frsum = 0;
      level->frcounter = 0;
    }
  }
}

static void stats_add(AspAecStats* st, float instant, float dtmp) { /* the repeated block of core:689-705 */
  st->instant = instant;
  if (dtmp > st->max) st->max = dtmp;
  if (dtmp < st->min) st->min = dtmp;
  st->counter++;
  st->sum += dtmp;
  st->average = st->sum / st->counter;
  if (dtmp > st->average) {
    st->hicounter++;
    st->hisum += dtmp;
    st->himean = st->hisum / st->hicounter;
  }
}

static void update_metrics(AspAecMetricsState* m, int echoState) { /* core:644-770 */
  const int subCountLen = 4, countLen = 50;
  const float actThresholdNoisy = 8.0f, actThresholdClean = 40.0f, safety = 0.99995f;
  const float noisyPower = 300000.0f;
  float dtmp, dtmp2, actThreshold, echo, suppressedEcho;
  if (echoState) m->stateCounter++;
  if (m->farlevel.frcounter == 0) {
    actThreshold = m->farlevel.minlevel < noisyPower ? actThresholdClean : actThresholdNoisy;
    if ((m->stateCounter > (0.5f * countLen * subCountLen)) && (m->farlevel.sfrcounter == 0) &&
        (m->farlevel.averagelevel > (actThreshold * m->farlevel.minlevel))) {
      echo = m->nearlevel.averagelevel - safety * m->nearlevel.minlevel;
      dtmp = 10 * (float)log10(m->farlevel.averagelevel / m->nearlevel.averagelevel + 1e-10f);
      dtmp2 = 10 * (float)log10(m->farlevel.averagelevel / echo + 1e-10f);
      (void)dtmp2;
      stats_add(&m->erl, dtmp, dtmp); /* ERL */
      dtmp = 10 * (float)log10(m->nearlevel.averagelevel / (2 * m->linoutlevel.averagelevel) + 1e-10f);
      suppressedEcho = 2 * (m->linoutlevel.averagelevel - safety * m->linoutlevel.minlevel);
      dtmp2 = 10 * (float)log10(echo / suppressedEcho + 1e-10f);
      stats_add(&m->aNlp, dtmp2, dtmp); /* A_NLP: instant takes dtmp2, the statistics dtmp (core:714-729) */
      suppressedEcho = 2 * (m->nlpoutlevel.averagelevel - safety * m->nlpoutlevel.minlevel);
      dtmp = 10 * (float)log10(m->nearlevel.averagelevel / (2 * m->nlpoutlevel.averagelevel) + 1e-10f);
      dtmp2 = 10 * (float)log10(echo / suppressedEcho + 1e-10f);
      dtmp = dtmp2;
      stats_add(&m->erle, dtmp, dtmp); /* ERLE */
    }
    m->stateCounter = 0;
  }
}

/* ------------------------------------------------------------ delay estimator
 * utility/delay_estimator.c (de:) and delay_estimator_wrapper.c (dw:), float path, robust validation on
 * (core:1534), history 125 blocks, near history 126 (max_lookahead = kHistorySizeBlocks), lookahead 15
 * (core:1356-1377). */
enum { DE_HIST = ASP_AEC_DELAY_HISTORY, DE_NEAR = ASP_AEC_DELAY_HISTORY + 1, DE_BAND_FIRST = 12, DE_BAND_LAST = 43 };
static const int32_t kMaxBitCountsQ9 = (32 << 9); /* de.h */

static void de_init(AspAecDelayState* d) { /* dw:175-191, 305-322; de:303-307, 476-498; core:1502-1516 */
  const int lookahead = d->lookahead, allowed = d->allowed_offset;
  memset(d, 0, sizeof *d);
  d->lookahead = lookahead; /* not touched by the Init functions */
  d->allowed_offset = allowed;
  for (int i = 0; i <= DE_HIST; ++i) d->mean_bit_counts[i] = (20 << 9);
  d->minimum_probability = kMaxBitCountsQ9;
  d->last_delay_probability = (int)kMaxBitCountsQ9;
  d->last_delay = -2;
  d->last_candidate_delay = -2;
  d->compare_delay = DE_HIST;
  d->previous_delay = -2;
  d->shift_offset = 5; /* kInitialShiftOffset, core:102 */
}

static int de_bitcount(uint32_t u32) { /* de:38-47 */
  uint32_t tmp = u32 - ((u32 >> 1) & 033333333333) - ((u32 >> 2) & 011111111111);
  tmp = ((tmp + (tmp >> 3)) & 030707070707);
  tmp = (tmp + (tmp >> 6));
  tmp = (tmp + (tmp >> 12) + (tmp >> 24)) & 077;
  return (int)tmp;
}

static void de_mean_fix(int32_t new_value, int factor, int32_t* mean_value) { /* de:672-684 */
  int32_t diff = new_value - *mean_value;
  if (diff < 0) {
    diff = -((-diff) >> factor);
  } else {
    diff = (diff >> factor);
  }
  *mean_value += diff;
}

static uint32_t de_binary_spectrum(const float* spectrum, float* threshold, int32_t* initialized) { /* dw:96-124 */
  const float kScale = 1 / 64.0;
  uint32_t out = 0;
  int i;
  if (!(*initialized)) {
    for (i = DE_BAND_FIRST; i <= DE_BAND_LAST; i++) {
      if (spectrum[i] > 0.0f) {
        threshold[i] = (spectrum[i] / 2);
        *initialized = 1;
      }
    }
  }
  for (i = DE_BAND_FIRST; i <= DE_BAND_LAST; i++) {
    threshold[i] += (spectrum[i] - threshold[i]) * kScale; /* MeanEstimatorFloat, dw:43-48 */
    if (spectrum[i] > threshold[i]) out |= (1u << (i - DE_BAND_FIRST));
  }
  return out;
}

static void de_add_binary_far(AspAecDelayState* d, uint32_t b) { /* WebRtc_AddBinaryFarSpectrum, de:356-369 */
  memmove(&d->binary_far_history[1], &d->binary_far_history[0], (DE_HIST - 1) * sizeof(uint32_t));
  d->binary_far_history[0] = b;
  memmove(&d->far_bit_counts[1], &d->far_bit_counts[0], (DE_HIST - 1) * sizeof(int32_t));
  d->far_bit_counts[0] = de_bitcount(b);
}

static void de_add_far(AspAecDelayState* d, const float* far_spectrum) { /* WebRtc_AddFarSpectrumFloat, dw:231-255 */
  de_add_binary_far(d, de_binary_spectrum(far_spectrum, d->mean_far_spectrum, &d->far_spectrum_initialized));
}

static void de_update_robust(AspAecDelayState* d, int candidate_delay, int32_t valley_depth_q14,
                             int32_t valley_level_q14) { /* de:90-146 */
  const float kQ14Scaling = 1.f / (1 << 14);
  const float valley_depth = valley_depth_q14 * kQ14Scaling;
  float decrease_in_last_set = valley_depth;
  const int max_hits_for_slow_change = (candidate_delay < d->last_delay) ? 10 : 1000;
  if (candidate_delay != d->last_candidate_delay) {
    d->candidate_hits = 0;
    d->last_candidate_delay = candidate_delay;
  }
  d->candidate_hits++;
  d->histogram[candidate_delay] += valley_depth;
  if (d->histogram[candidate_delay] > 3000.f) d->histogram[candidate_delay] = 3000.f;
  if (d->candidate_hits < max_hits_for_slow_change)
    decrease_in_last_set = (d->mean_bit_counts[d->compare_delay] - valley_level_q14) * kQ14Scaling;
  for (int i = 0; i < DE_HIST; ++i) {
    const int is_in_last_set = (i >= d->last_delay - 2) && (i <= d->last_delay + 1) && (i != candidate_delay);
    const int is_in_candidate_set = (i >= candidate_delay - 2) && (i <= candidate_delay + 1);
    d->histogram[i] -= decrease_in_last_set * is_in_last_set + valley_depth * (!is_in_last_set && !is_in_candidate_set);
    if (d->histogram[i] < 0) d->histogram[i] = 0;
  }
}

static int de_histogram_valid(const AspAecDelayState* d, int candidate_delay) { /* de:173-214 */
  float fraction = 1.f;
  float histogram_threshold = d->histogram[d->compare_delay];
  const int delay_difference = candidate_delay - d->last_delay;
  if (delay_difference > d->allowed_offset) {
    fraction = 1.f - 0.05f * (delay_difference - d->allowed_offset);
    fraction = (fraction > 0.5f ? fraction : 0.5f);
  } else if (delay_difference < 0) {
    fraction = 0.25f - 0.05f * delay_difference;
    fraction = (fraction > 1.f ? 1.f : fraction);
  }
  histogram_threshold *= fraction;
  histogram_threshold = (histogram_threshold > 1.5f ? histogram_threshold : 1.5f);
  return (d->histogram[candidate_delay] >= histogram_threshold) && (d->candidate_hits > 10);
}

static int de_process_binary(AspAecDelayState* d, uint32_t binary_near) { /* WebRtc_ProcessBinarySpectrum, de:513-644 */
  int i, candidate_delay = -1, valid_candidate;
  int32_t value_best_candidate = kMaxBitCountsQ9, value_worst_candidate = 0, valley_depth;
  memmove(&d->binary_near_history[1], &d->binary_near_history[0], (DE_NEAR - 1) * sizeof(uint32_t));
  d->binary_near_history[0] = binary_near;
  binary_near = d->binary_near_history[d->lookahead];
  for (i = 0; i < DE_HIST; i++) d->bit_counts[i] = (int32_t)de_bitcount(binary_near ^ d->binary_far_history[i]);
  for (i = 0; i < DE_HIST; i++) {
    const int32_t bit_count = (d->bit_counts[i] << 9);
    if (d->far_bit_counts[i] > 0) {
      int shifts = 13;                          /* kShiftsAtZero */
      shifts -= (3 * d->far_bit_counts[i]) >> 4; /* kShiftsLinearSlope */
      de_mean_fix(bit_count, shifts, &d->mean_bit_counts[i]);
    }
  }
  for (i = 0; i < DE_HIST; i++) {
    if (d->mean_bit_counts[i] < value_best_candidate) {
      value_best_candidate = d->mean_bit_counts[i];
      candidate_delay = i;
    }
    if (d->mean_bit_counts[i] > value_worst_candidate) value_worst_candidate = d->mean_bit_counts[i];
  }
  valley_depth = value_worst_candidate - value_best_candidate;
  if ((d->minimum_probability > 8704) && (valley_depth > 2816)) { /* kProbabilityLowerLimit, kProbabilityMinSpread */
    int32_t threshold = value_best_candidate + 1024;                /* kProbabilityOffset */
    if (threshold < 8704) threshold = 8704;
    if (d->minimum_probability > threshold) d->minimum_probability = threshold;
  }
  d->last_delay_probability++;
  valid_candidate = ((valley_depth > 1024) && ((value_best_candidate < d->minimum_probability) ||
                                               (value_best_candidate < d->last_delay_probability)));
  { /* robust validation, de:610-618 and 236-258 */
    int is_histogram_valid, is_robust;
    de_update_robust(d, candidate_delay, valley_depth, value_best_candidate);
    is_histogram_valid = de_histogram_valid(d, candidate_delay);
    is_robust = (d->last_delay < 0) && (valid_candidate || is_histogram_valid);
    is_robust |= valid_candidate && is_histogram_valid;
    is_robust |= is_histogram_valid && (d->histogram[candidate_delay] > d->last_delay_histogram);
    valid_candidate = is_robust;
  }
  if (valid_candidate) {
    if (candidate_delay != d->last_delay) {
      d->last_delay_histogram = (d->histogram[candidate_delay] > 250.f ? 250.f : d->histogram[candidate_delay]);
      if (d->histogram[candidate_delay] < d->histogram[d->compare_delay])
        d->histogram[d->compare_delay] = d->histogram[candidate_delay];
    }
    d->last_delay = candidate_delay;
    if (value_best_candidate < d->last_delay_probability) d->last_delay_probability = value_best_candidate;
    d->compare_delay = d->last_delay;
  }
  return d->last_delay;
}

static int de_process(AspAecDelayState* d, const float* near_spectrum) { /* WebRtc_DelayEstimatorProcessFloat, dw:446-469 */
  return de_process_binary(d, de_binary_spectrum(near_spectrum, d->mean_near_spectrum, &d->near_spectrum_initialized));
}

static float de_quality(const AspAecDelayState* d) { return d->histogram[d->compare_delay] / 3000.f; } /* de:655-658 */

static void de_soft_reset(AspAecDelayState* d, int delay_shift) { /* de:309-339, 500-511 */
  const int abs_shift = abs(delay_shift);
  const int shift_size = DE_HIST - abs_shift;
  int dest_index = 0, src_index = 0, padding_index = 0;
  d->lookahead -= delay_shift; /* WebRtc_SoftResetBinaryDelayEstimator */
  if (d->lookahead < 0) d->lookahead = 0;
  if (d->lookahead > DE_NEAR - 1) d->lookahead = DE_NEAR - 1;
  if (delay_shift == 0) return; /* ...Farend */
  if (delay_shift > 0) {
    dest_index = abs_shift;
  } else {
    src_index = abs_shift;
    padding_index = shift_size;
  }
  memmove(&d->binary_far_history[dest_index], &d->binary_far_history[src_index], sizeof(uint32_t) * shift_size);
  memset(&d->binary_far_history[padding_index], 0, sizeof(uint32_t) * abs_shift);
  memmove(&d->far_bit_counts[dest_index], &d->far_bit_counts[src_index], sizeof(int32_t) * shift_size);
  memset(&d->far_bit_counts[padding_index], 0, sizeof(int32_t) * abs_shift);
}

static void init_core(AspAecOracle* o, int sampFreq) { /* WebRtcAec_InitAec, core:1460-1615 */
  AspAecState* s = &o->st;
  if (sampFreq == 8000) {
    o->normal_mu = 0.6f;
    o->normal_error_threshold = 2e-6f;
  } else {
    o->normal_mu = 0.5f;
    o->normal_error_threshold = 1.5e-6f;
  }
  rp_init(&o->near_pos, FRBUF_LEN);
  rp_init(&o->out_pos, FRBUF_LEN);
  rp_init(&o->far_pos, FAR_SLOTS);
  memset(o->nearfr, 0, sizeof o->nearfr);
  memset(o->outfr, 0, sizeof o->outfr);
  memset(o->nearfrH, 0, sizeof o->nearfrH);
  memset(o->outfrH, 0, sizeof o->outfrH);
  memset(o->far, 0, sizeof o->far);
  memset(o->farw, 0, sizeof o->farw);
  o->system_delay = 0;
  o->delay_logging = 0;
  o->nlp_mode = 1;
  o->num_bands = sampFreq > 16000 ? sampFreq / 16000 : 1; /* core:1466-1473 */
  o->mult = o->num_bands > 1 ? (short)sampFreq / 16000 : (short)sampFreq / 8000; /* core:1541-1545 */
  o->core_knownDelay = 0;
  memset(s, 0, sizeof *s);
  for (int i = 0; i < PART_LEN1; i++) s->dMinPow[i] = 1.0e6f;
  for (int i = 0; i < PART_LEN1; i++) s->sd[i] = 1;
  for (int i = 0; i < PART_LEN1; i++) s->sx[i] = 1;
  s->hNlFbMin = 1;
  s->hNlFbLocalMin = 1;
  s->hNlXdAvgMin = 1;
  s->hNlNewMin = 0;
  s->hNlMinCtr = 0;
  s->overDrive = 2;
  s->overDriveSm = 2;
  s->delayIdx = 0;
  s->stNearState = 0;
  s->echoState = 0;
  s->divergeState = 0;
  s->seed = 777;
  s->delayEstCtr = 0;
  s->xfBufBlockPos = 0;
  s->noiseEstCtr = 0;
  o->metricsMode = 0;
  init_metrics(&o->met);
  o->blocks_processed = 0;
  o->extended_filter_enabled = 0; /* core:1522-1523 */
  o->num_partitions = NPART_NORMAL;
  o->reported_delay_enabled = 1;  /* core:1517-1521 (not Android) */
  o->de.allowed_offset = o->num_partitions / 2; /* core:1529 */
  de_init(&o->de);
}

/* WebRtcAec_enable_delay_correction / _delay_correction_enabled, core:1876-1885 (reached through
 * WebRtcAec_aec_core(handle)). */
/* The estimator on its own (the seam utility/delay_estimator_unittest.cc tests): robust validation on, history 125. */
void asp_de_oracle_init(AspAecDelayState* d, int lookahead, int allowed_offset) {
  d->lookahead = lookahead;
  d->allowed_offset = allowed_offset;
  de_init(d);
}
void asp_de_oracle_add_binary_far(AspAecDelayState* d, uint32_t binary_far) { de_add_binary_far(d, binary_far); }
int asp_de_oracle_process_binary(AspAecDelayState* d, uint32_t binary_near) { return de_process_binary(d, binary_near); }
void asp_de_oracle_add_far(AspAecDelayState* d, const float* far_spectrum) { de_add_far(d, far_spectrum); }
int asp_de_oracle_process(AspAecDelayState* d, const float* near_spectrum) { return de_process(d, near_spectrum); }
float asp_de_oracle_quality(const AspAecDelayState* d) { return de_quality(d); }

void asp_aec_oracle_enable_delay_correction(AspAecOracle* o, int enable) {
  o->extended_filter_enabled = enable;
  o->num_partitions = enable ? NPART_MAX : NPART_NORMAL;
  o->de.allowed_offset = o->num_partitions / 2; /* core:1880 */
}
/* WebRtcAec_enable_reported_delay / _reported_delay_enabled, core:1868-1874: 0 = the delay-agnostic mode */
void asp_aec_oracle_enable_reported_delay(AspAecOracle* o, int enable) { o->reported_delay_enabled = enable; }
int asp_aec_oracle_reported_delay_enabled(const AspAecOracle* o) { return o->reported_delay_enabled; }
int asp_aec_oracle_delay_correction_enabled(const AspAecOracle* o) { return o->extended_filter_enabled; }

int asp_aec_oracle_set_config(AspAecOracle* o, AecConfig config) { /* ec:410-438 */
  if (o->initFlag != kInitCheck) {
    o->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (config.skewMode != kAecFalse && config.skewMode != kAecTrue) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  o->skewMode = config.skewMode;
  if (config.nlpMode != kAecNlpConservative && config.nlpMode != kAecNlpModerate &&
      config.nlpMode != kAecNlpAggressive) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (config.metricsMode != kAecFalse && config.metricsMode != kAecTrue) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (config.delay_logging != kAecFalse && config.delay_logging != kAecTrue) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  o->nlp_mode = config.nlpMode; /* WebRtcAec_SetConfigCore, core:1844-1862 */
  o->metricsMode = config.metricsMode;
  if (o->metricsMode) init_metrics(&o->met);
  o->delay_logging = config.delay_logging;
  if (o->delay_logging) memset(o->de.delay_histogram, 0, sizeof o->de.delay_histogram);
  return 0;
}

int asp_aec_oracle_init(AspAecOracle* o, int32_t sampFreq, int32_t scSampFreq) { /* ec:196-276 */
  AecConfig cfg;
  if (sampFreq != 8000 && sampFreq != 16000 && sampFreq != 32000 && sampFreq != 48000) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (sampFreq > 32000) { /* the reference's own mult is broken at 48 kHz (core:1541-1543) */
    o->lastError = AEC_UNSUPPORTED_FUNCTION_ERROR;
    return -1;
  }
  o->sampFreq = sampFreq;
  if (scSampFreq < 1 || scSampFreq > 96000) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  o->scSampFreq = scSampFreq;
  init_core(o, sampFreq);
  memset(o->rs_buffer, 0, sizeof o->rs_buffer); /* WebRtcAec_InitResampler(resampler, scSampFreq), ec:221; rs:55-66 */
  o->rs_position = 0.0;
  o->rs_deviceSampleRateHz = scSampFreq;
  memset(o->rs_skewData, 0, sizeof o->rs_skewData);
  o->rs_skewDataIndex = 0;
  o->rs_skewEstimate = 0.0;
  rp_init(&o->pre_pos, PRE_LEN);
  memset(o->pre, 0, sizeof o->pre);
  rp_move_read(&o->pre_pos, -PART_LEN); /* start overlap, ec:226 */
  o->initFlag = kInitCheck;
  o->splitSampFreq = sampFreq == 32000 ? 16000 : sampFreq; /* ec:231-235 */
  o->rate_factor = o->splitSampFreq / 8000;
  o->sampFactor = (o->scSampFreq * 1.0f) / o->splitSampFreq; /* ec:238 */
  o->sum = 0;
  o->counter = 0;
  o->checkBuffSize = 1;
  o->firstVal = 0;
  o->startup_phase = o->reported_delay_enabled; /* ec:247 (1: InitAec has just set it) */
  o->bufSizeStart = 0;
  o->checkBufSizeCtr = 0;
  o->msInSndCardBuf = 0;
  o->filtDelay = -1;
  o->timeForDelayChange = 0;
  o->knownDelay = 0;
  o->lastDelayDiff = 0;
  o->skewFrCtr = 0; /* ec:256-259 */
  o->resample = kAecFalse;
  o->highSkewCtr = 0;
  o->skew = 0;
  o->farend_started = 0;
  cfg.nlpMode = kAecNlpModerate;
  cfg.skewMode = kAecFalse;
  cfg.metricsMode = kAecFalse;
  cfg.delay_logging = kAecFalse;
  if (asp_aec_oracle_set_config(o, cfg) == -1) {
    o->lastError = AEC_UNSPECIFIED_ERROR;
    return -1;
  }
  return 0;
}

/* ------------------------------------------------------------- block maths */
static void window128(float* x) { /* TimeToFrequency window branch, core:779-784 */
  for (int i = 0; i < PART_LEN; i++) {
    x[i] *= g_hann[i];
    x[PART_LEN + i] *= g_hann[PART_LEN - i];
  }
}

static void unpack_spectrum(const float* t, float f[2][PART_LEN1]) { /* core:786-795 */
  f[1][0] = 0;
  f[1][PART_LEN] = 0;
  f[0][0] = t[0];
  f[0][PART_LEN] = t[1];
  for (int i = 1; i < PART_LEN; i++) {
    f[0][i] = t[2 * i];
    f[1][i] = t[2 * i + 1];
  }
}

static void buffer_farend_partition(AspAecOracle* o, const float* farend) { /* core:1618-1635 */
  float fft[PART_LEN2];
  float xf[2][PART_LEN1];
  if (rp_avail_write(&o->far_pos) < 1) far_move_read(o, 1);
  memcpy(fft, farend, sizeof fft);
  asp_aec_oracle_rdft128(fft, 1);
  unpack_spectrum(fft, xf);
  {
    RingPos keep = o->far_pos; /* far_buf and far_buf_windowed move in lock-step */
    ring_write(&keep, &o->far[0][0], 2 * PART_LEN1, &xf[0][0], 1);
  }
  memcpy(fft, farend, sizeof fft);
  window128(fft);
  asp_aec_oracle_rdft128(fft, 1);
  unpack_spectrum(fft, xf);
  ring_write(&o->far_pos, &o->farw[0][0], 2 * PART_LEN1, &xf[0][0], 1);
}

static int partition_delay(const AspAecState* s, int num_partitions) { /* core:294-318 */
  float wfEnMax = 0;
  int delay = 0;
  for (int i = 0; i < num_partitions; i++) {
    const int pos = i * PART_LEN1;
    float wfEn = 0;
    for (int j = 0; j < PART_LEN1; j++)
      wfEn += s->wfBuf[0][pos + j] * s->wfBuf[0][pos + j] + s->wfBuf[1][pos + j] * s->wfBuf[1][pos + j];
    if (wfEn > wfEnMax) {
      wfEnMax = wfEn;
      delay = i;
    }
  }
  return delay;
}

static void sort_floats(float* v, int n) { /* qsort + CmpFloat, core:140-145,956 */
  for (int i = 1; i < n; ++i) {
    const float x = v[i];
    int j = i - 1;
    while (j >= 0 && v[j] > x) {
      v[j + 1] = v[j];
      --j;
    }
    v[j + 1] = x;
  }
}

static void nonlinear_processing(AspAecOracle* o, float* output, float* outputH,
                                 const float* noisePow) { /* core:852-1082 */
  AspAecState* s = &o->st;
  float efw[2][PART_LEN1], xfw[2][PART_LEN1], dfw[2][PART_LEN1];
  float fft[PART_LEN2];
  float cohde[PART_LEN1], cohxd[PART_LEN1], hNl[PART_LEN1];
  float hNlPref[24];
  float hNlDeAvg, hNlXdAvg, hNlFb = 0, hNlFbLow = 0;
  const float prefBandQuant = 0.75f, prefBandQuantLow = 0.5f;
  const int prefBandSize = 24 / o->mult, minPrefBand = 4 / o->mult;
  static const float kNormalMinOverDrive[3] = {1.0f, 2.0f, 5.0f};    /* core:110 */
  static const float kExtendedMinOverDrive[3] = {3.0f, 6.0f, 15.0f}; /* core:109 */
  const float* kMinOverDrive = o->extended_filter_enabled ? kExtendedMinOverDrive : kNormalMinOverDrive; /* core:872-874 */
  static const float kTargetSupp[3] = {-6.9f, -11.5f, -18.4f}; /* core:104 */
  static const float kCoef[2][2] = {{0.9f, 0.1f}, {0.93f, 0.07f}};    /* kNormalSmoothingCoefficients, core:113-114 */
  static const float kCoefExt[2][2] = {{0.9f, 0.1f}, {0.92f, 0.08f}}; /* kExtendedSmoothingCoefficients, core:111-112 */
  const float* gc = (o->extended_filter_enabled ? kCoefExt : kCoef)[o->mult - 1]; /* core:337-339 */
  const int delayEstInterval = 10 * o->mult;
  float* xfw_raw = &s->xfwBuf[0][0];
  float sdSum = 0, seSum = 0;
  int i;

  s->delayEstCtr++;
  if (s->delayEstCtr == delayEstInterval) s->delayEstCtr = 0;

  /* partition 0 of xfwBuf was filled by process_block (core:886-891) */

  /* SubbandCoherence, core:411-449 */
  if (s->delayEstCtr == 0) s->delayIdx = partition_delay(s, o->num_partitions);
  memcpy(xfw, xfw_raw + (size_t)s->delayIdx * 2 * PART_LEN1, sizeof xfw);
  memcpy(fft, s->dBuf, sizeof fft);
  window128(fft);
  asp_aec_oracle_rdft128(fft, 1);
  dfw[0][0] = fft[0]; /* StoreAsComplex, core:398-409 */
  dfw[1][0] = 0;
  for (i = 1; i < PART_LEN; i++) {
    dfw[0][i] = fft[2 * i];
    dfw[1][i] = fft[2 * i + 1];
  }
  dfw[0][PART_LEN] = fft[1];
  dfw[1][PART_LEN] = 0;
  memcpy(fft, s->eBuf, sizeof fft);
  window128(fft);
  asp_aec_oracle_rdft128(fft, 1);
  efw[0][0] = fft[0];
  efw[1][0] = 0;
  for (i = 1; i < PART_LEN; i++) {
    efw[0][i] = fft[2 * i];
    efw[1][i] = fft[2 * i + 1];
  }
  efw[0][PART_LEN] = fft[1];
  efw[1][PART_LEN] = 0;

  /* SmoothedPSD, core:332-385 */
  for (i = 0; i < PART_LEN1; i++) {
    float xx;
    s->sd[i] = gc[0] * s->sd[i] + gc[1] * (dfw[0][i] * dfw[0][i] + dfw[1][i] * dfw[1][i]);
    s->se[i] = gc[0] * s->se[i] + gc[1] * (efw[0][i] * efw[0][i] + efw[1][i] * efw[1][i]);
    xx = xfw[0][i] * xfw[0][i] + xfw[1][i] * xfw[1][i];
    s->sx[i] = gc[0] * s->sx[i] + gc[1] * (xx > 15 ? xx : 15); /* WebRtcAec_kMinFarendPSD */
    s->sde[i][0] = gc[0] * s->sde[i][0] + gc[1] * (dfw[0][i] * efw[0][i] + dfw[1][i] * efw[1][i]);
    s->sde[i][1] = gc[0] * s->sde[i][1] + gc[1] * (dfw[0][i] * efw[1][i] - dfw[1][i] * efw[0][i]);
    s->sxd[i][0] = gc[0] * s->sxd[i][0] + gc[1] * (dfw[0][i] * xfw[0][i] + dfw[1][i] * xfw[1][i]);
    s->sxd[i][1] = gc[0] * s->sxd[i][1] + gc[1] * (dfw[0][i] * xfw[1][i] - dfw[1][i] * xfw[0][i]);
    sdSum += s->sd[i];
    seSum += s->se[i];
  }
  s->divergeState = (s->divergeState ? 1.05f : 1.0f) * seSum > sdSum;
  if (s->divergeState) memcpy(efw, dfw, sizeof efw);
  if (!o->extended_filter_enabled && seSum > (19.95f * sdSum)) memset(s->wfBuf, 0, sizeof s->wfBuf); /* core:383 */

  for (i = 0; i < PART_LEN1; i++) { /* core:439-448 */
    cohde[i] = (s->sde[i][0] * s->sde[i][0] + s->sde[i][1] * s->sde[i][1]) / (s->sd[i] * s->se[i] + 1e-10f);
    cohxd[i] = (s->sxd[i][0] * s->sxd[i][0] + s->sxd[i][1] * s->sxd[i][1]) / (s->sx[i] * s->sd[i] + 1e-10f);
  }

  hNlXdAvg = 0; /* core:895-907 */
  for (i = minPrefBand; i < prefBandSize + minPrefBand; i++) hNlXdAvg += cohxd[i];
  hNlXdAvg /= prefBandSize;
  hNlXdAvg = 1 - hNlXdAvg;
  hNlDeAvg = 0;
  for (i = minPrefBand; i < prefBandSize + minPrefBand; i++) hNlDeAvg += cohde[i];
  hNlDeAvg /= prefBandSize;

  if (hNlXdAvg < 0.75f && hNlXdAvg < s->hNlXdAvgMin) s->hNlXdAvgMin = hNlXdAvg;
  if (hNlDeAvg > 0.98f && hNlXdAvg > 0.9f) {
    s->stNearState = 1;
  } else if (hNlDeAvg < 0.95f || hNlXdAvg < 0.8f) {
    s->stNearState = 0;
  }

  if (s->hNlXdAvgMin == 1) { /* core:919-935 */
    s->echoState = 0;
    s->overDrive = kMinOverDrive[o->nlp_mode];
    if (s->stNearState == 1) {
      memcpy(hNl, cohde, sizeof hNl);
      hNlFb = hNlDeAvg;
      hNlFbLow = hNlDeAvg;
    } else {
      for (i = 0; i < PART_LEN1; i++) hNl[i] = 1 - cohxd[i];
      hNlFb = hNlXdAvg;
      hNlFbLow = hNlXdAvg;
    }
  } else { /* core:936-960 */
    if (s->stNearState == 1) {
      s->echoState = 0;
      memcpy(hNl, cohde, sizeof hNl);
      hNlFb = hNlDeAvg;
      hNlFbLow = hNlDeAvg;
    } else {
      s->echoState = 1;
      for (i = 0; i < PART_LEN1; i++) {
        const float a = cohde[i], b = 1 - cohxd[i];
        hNl[i] = a < b ? a : b;
      }
      memcpy(hNlPref, &hNl[minPrefBand], sizeof(float) * prefBandSize);
      sort_floats(hNlPref, prefBandSize);
      hNlFb = hNlPref[(int)floor(prefBandQuant * (prefBandSize - 1))];
      hNlFbLow = hNlPref[(int)floor(prefBandQuantLow * (prefBandSize - 1))];
    }
  }

  /* core:962-992 */
  if (hNlFbLow < 0.6f && hNlFbLow < s->hNlFbLocalMin) {
    s->hNlFbLocalMin = hNlFbLow;
    s->hNlFbMin = hNlFbLow;
    s->hNlNewMin = 1;
    s->hNlMinCtr = 0;
  }
  {
    const float a = s->hNlFbLocalMin + 0.0008f / o->mult;
    const float b = s->hNlXdAvgMin + 0.0006f / o->mult;
    s->hNlFbLocalMin = a < 1 ? a : 1;
    s->hNlXdAvgMin = b < 1 ? b : 1;
  }
  if (s->hNlNewMin == 1) s->hNlMinCtr++;
  if (s->hNlMinCtr == 2) {
    float v;
    s->hNlNewMin = 0;
    s->hNlMinCtr = 0;
    v = kTargetSupp[o->nlp_mode] / ((float)log(s->hNlFbMin + 1e-10f) + 1e-10f);
    s->overDrive = v > kMinOverDrive[o->nlp_mode] ? v : kMinOverDrive[o->nlp_mode];
  }
  if (s->overDrive < s->overDriveSm) {
    s->overDriveSm = 0.99f * s->overDriveSm + 0.01f * s->overDrive;
  } else {
    s->overDriveSm = 0.9f * s->overDriveSm + 0.1f * s->overDrive;
  }

  /* OverdriveAndSuppress, core:271-292 */
  for (i = 0; i < PART_LEN1; i++) {
    if (hNl[i] > hNlFb) hNl[i] = g_weight[i] * hNlFb + (1 - g_weight[i]) * hNl[i];
    hNl[i] = powf(hNl[i], s->overDriveSm * g_odrive[i]);
    efw[0][i] *= hNl[i];
    efw[1][i] *= hNl[i];
    efw[1][i] *= -1;
  }

  /* ComfortNoise, core:461-545 */
  float cnH[PART_LEN1][2];
  memset(cnH, 0, sizeof cnH);
  {
    float rnd[PART_LEN];
    float u[PART_LEN1][2];
    const float pi2 = 6.28318530717959f;
    for (i = 0; i < PART_LEN; i++) { /* WebRtcSpl_RandUArray, randomization_functions.c:93-115 */
      s->seed = (s->seed * 69069u + 1u) & 0x7fffffffu;
      rnd[i] = ((float)(int16_t)(s->seed >> 16)) / 32768;
    }
    u[0][0] = 0;
    u[0][1] = 0;
    for (i = 1; i < PART_LEN1; i++) {
      const float tmp = pi2 * rnd[i - 1];
      const float noise = sqrtf(noisePow[i]);
      u[i][0] = noise * cosf(tmp);
      u[i][1] = -noise * sinf(tmp);
    }
    u[PART_LEN][1] = 0;
    for (i = 0; i < PART_LEN1; i++) {
      const float r = 1 - hNl[i] * hNl[i];
      const float tmp = sqrtf(r > 0 ? r : 0);
      efw[0][i] += tmp * u[i][0];
      efw[1][i] += tmp * u[i][1];
    }
    if (o->num_bands > 1) { /* H band comfort noise, core:501-545 */
      float noiseAvg = 0.0, tmpAvg = 0.0;
      int num = 0;
      for (i = PART_LEN1 >> 1; i < PART_LEN1; i++) {
        num++;
        noiseAvg += sqrtf(noisePow[i]);
      }
      noiseAvg /= (float)num;
      num = 0;
      for (i = PART_LEN1 >> 1; i < PART_LEN1; i++) {
        const float r = 1 - hNl[i] * hNl[i];
        num++;
        tmpAvg += sqrtf(r > 0 ? r : 0);
      }
      tmpAvg /= (float)num;
      u[0][0] = 0;
      u[0][1] = 0;
      for (i = 1; i < PART_LEN1; i++) {
        const float tmp = pi2 * rnd[i - 1];
        u[i][0] = noiseAvg * (float)cos(tmp);
        u[i][1] = -noiseAvg * (float)sin(tmp);
      }
      u[PART_LEN][1] = 0;
      for (i = 0; i < PART_LEN1; i++) {
        cnH[i][0] = tmpAvg * u[i][0];
        cnH[i][1] = tmpAvg * u[i][1];
      }
    }
  }

  if (o->metricsMode == 1) update_level(&o->met.nlpoutlevel, efw); /* core:1000-1006 */

  /* inverse error fft, overlap-add, saturate: core:1006-1030 */
  fft[0] = efw[0][0];
  fft[1] = efw[0][PART_LEN];
  for (i = 1; i < PART_LEN; i++) {
    fft[2 * i] = efw[0][i];
    fft[2 * i + 1] = -efw[1][i];
  }
  asp_aec_oracle_rdft128(fft, -1);
  {
    const float scale = 2.0f / PART_LEN2;
    for (i = 0; i < PART_LEN; i++) {
      float v;
      fft[i] *= scale;
      fft[i] = fft[i] * g_hann[i] + s->outBuf[i];
      fft[PART_LEN + i] *= scale;
      s->outBuf[i] = fft[PART_LEN + i] * g_hann[PART_LEN - i];
      v = fft[i];
      output[i] = v > 32767 ? 32767 : (v < -32768 ? -32768 : v); /* WEBRTC_SPL_SAT */
    }
  }

  if (o->num_bands > 1) { /* core:1032-1067 */
    float nlpGainHband = (float)0.0;
    for (i = PART_LEN / 2; i < PART_LEN1 - 1; i++) nlpGainHband += hNl[i]; /* GetHighbandGain, core:451-459 */
    nlpGainHband /= (float)(PART_LEN1 - 1 - PART_LEN / 2);
    fft[0] = cnH[0][0];
    fft[1] = cnH[PART_LEN][0];
    for (i = 1; i < PART_LEN; i++) {
      fft[2 * i] = cnH[i][0];
      fft[2 * i + 1] = cnH[i][1];
    }
    asp_aec_oracle_rdft128(fft, -1);
    {
      const float scale = 2.0f / PART_LEN2;
      for (i = 0; i < PART_LEN; i++) {
        float dtmp = s->dBufH[i];
        dtmp = dtmp * nlpGainHband;
        fft[i] *= scale;
        dtmp += (float)0.4 * fft[i]; /* cnScaleHband, core:45-46 */
        outputH[i] = dtmp > 32767 ? 32767 : (dtmp < -32768 ? -32768 : dtmp);
      }
    }
    memcpy(s->dBufH, s->dBufH + PART_LEN, sizeof(float) * PART_LEN); /* core:1075-1077 */
  }
  /* core:1069-1081 */
  memcpy(s->dBuf, s->dBuf + PART_LEN, sizeof(float) * PART_LEN);
  memcpy(s->eBuf, s->eBuf + PART_LEN, sizeof(float) * PART_LEN);
  /* the whole 32-partition array moves whatever the filter length is (core:1079-1081: sizeof(aec->xfwBuf)) */
  memmove(xfw_raw + 2 * PART_LEN1, xfw_raw, sizeof(float) * 2 * PART_LEN1 * (NPART_MAX - 1));
}

static void process_block(AspAecOracle* o) { /* core:1084-1287 */
  AspAecState* s = &o->st;
  float fft[PART_LEN2];
  float xf[2][PART_LEN1], yf[2][PART_LEN1], ef[2][PART_LEN1], df[2][PART_LEN1];
  float nearend[PART_LEN], y[PART_LEN], e[PART_LEN], output[PART_LEN];
  const float gPow[2] = {0.9f, 0.1f};
  const int noiseInitBlocks = 500 * o->mult;
  const float step = 0.1f, ramp = 1.0002f;
  const float gInitNoise[2] = {0.999f, 0.001f};
  const float* noisePow = s->dMinPow;
  int i;

  if (o->num_bands > 1) { /* core:1117-1123: the high band's ring moves with the low band's */
    RingPos keep = o->near_pos;
    float nearH[PART_LEN];
    ring_read(&keep, o->nearfrH, 1, nearH, PART_LEN);
    memcpy(s->dBufH + PART_LEN, nearH, sizeof nearH);
  }
  ring_read(&o->near_pos, o->nearfr, 1, nearend, PART_LEN);
  memcpy(s->dBuf + PART_LEN, nearend, sizeof nearend);

  /* one element of far_buf now and the same element of far_buf_windowed in the NLP
   * (core:1137, 888): both rings move in lock-step, so read both here */
  {
    RingPos keep = o->far_pos;
    ring_read(&keep, &o->far[0][0], 2 * PART_LEN1, &xf[0][0], 1);
  }
  {
    float xfw_now[2 * PART_LEN1];
    ring_read(&o->far_pos, &o->farw[0][0], 2 * PART_LEN1, xfw_now, 1);
    memcpy(&s->xfwBuf[0][0], xfw_now, sizeof xfw_now); /* core:891 */
  }

  memcpy(fft, s->dBuf, sizeof fft); /* near fft, core:1140-1141 */
  asp_aec_oracle_rdft128(fft, 1);
  unpack_spectrum(fft, df);

  float abs_far_spectrum[PART_LEN1], abs_near_spectrum[PART_LEN1];
  for (i = 0; i < PART_LEN1; i++) { /* core:1144-1156 */
    const float far_spectrum = (xf[0][i] * xf[0][i]) + (xf[1][i] * xf[1][i]);
    const float near_spectrum = df[0][i] * df[0][i] + df[1][i] * df[1][i];
    s->xPow[i] = gPow[0] * s->xPow[i] + gPow[1] * o->num_partitions * far_spectrum;
    s->dPow[i] = gPow[0] * s->dPow[i] + gPow[1] * near_spectrum;
    abs_far_spectrum[i] = sqrtf(far_spectrum);
    abs_near_spectrum[i] = sqrtf(near_spectrum);
  }
  if (s->noiseEstCtr > 50) { /* core:1159-1168 */
    for (i = 0; i < PART_LEN1; i++) {
      if (s->dPow[i] < s->dMinPow[i]) {
        s->dMinPow[i] = (s->dPow[i] + step * (s->dMinPow[i] - s->dPow[i])) * ramp;
      } else {
        s->dMinPow[i] *= ramp;
      }
    }
  }
  if (s->noiseEstCtr < noiseInitBlocks) { /* core:1172-1186 */
    noisePow = s->dInitMinPow;
    s->noiseEstCtr++;
    for (i = 0; i < PART_LEN1; i++) {
      if (s->dMinPow[i] > s->dInitMinPow[i]) {
        s->dInitMinPow[i] = gInitNoise[0] * s->dInitMinPow[i] + gInitNoise[1] * s->dMinPow[i];
      } else {
        s->dInitMinPow[i] = s->dMinPow[i];
      }
    }
  }
  if (o->delay_logging) { /* block-wise delay estimation, core:1191-1203 */
    de_add_far(&o->de, abs_far_spectrum);
    {
      const int delay_estimate = de_process(&o->de, abs_near_spectrum);
      if (delay_estimate >= 0) o->de.delay_histogram[delay_estimate]++;
    }
  }

  s->xfBufBlockPos--; /* core:1203-1214 */
  if (s->xfBufBlockPos == -1) s->xfBufBlockPos = o->num_partitions - 1;
  memcpy(s->xfBuf[0] + s->xfBufBlockPos * PART_LEN1, xf[0], sizeof(float) * PART_LEN1);
  memcpy(s->xfBuf[1] + s->xfBufBlockPos * PART_LEN1, xf[1], sizeof(float) * PART_LEN1);

  memset(yf, 0, sizeof yf);
  for (i = 0; i < o->num_partitions; i++) { /* FilterFar, core:147-169 */
    int xPos = (i + s->xfBufBlockPos) * PART_LEN1;
    const int pos = i * PART_LEN1;
    if (i + s->xfBufBlockPos >= o->num_partitions) xPos -= o->num_partitions * PART_LEN1;
    for (int j = 0; j < PART_LEN1; j++) {
      yf[0][j] += s->xfBuf[0][xPos + j] * s->wfBuf[0][pos + j] - s->xfBuf[1][xPos + j] * s->wfBuf[1][pos + j];
      yf[1][j] += s->xfBuf[0][xPos + j] * s->wfBuf[1][pos + j] + s->xfBuf[1][xPos + j] * s->wfBuf[0][pos + j];
    }
  }

  fft[0] = yf[0][0]; /* core:1222-1238 */
  fft[1] = yf[0][PART_LEN];
  for (i = 1; i < PART_LEN; i++) {
    fft[2 * i] = yf[0][i];
    fft[2 * i + 1] = yf[1][i];
  }
  asp_aec_oracle_rdft128(fft, -1);
  {
    const float scale = 2.0f / PART_LEN2;
    for (i = 0; i < PART_LEN; i++) y[i] = fft[PART_LEN + i] * scale;
  }
  for (i = 0; i < PART_LEN; i++) e[i] = nearend[i] - y[i];

  memcpy(s->eBuf + PART_LEN, e, sizeof e); /* error fft, core:1241-1254 */
  memset(fft, 0, sizeof(float) * PART_LEN);
  memcpy(fft + PART_LEN, e, sizeof e);
  asp_aec_oracle_rdft128(fft, 1);
  unpack_spectrum(fft, ef);

  if (o->metricsMode == 1) update_level(&o->met.linoutlevel, ef); /* core:1258-1263 */

  { /* ScaleErrorSignal, core:171-193 */
    /* core:172-175, core_internal:39-40 */
    const float mu = o->extended_filter_enabled ? 0.4f : o->normal_mu;
    const float error_threshold = o->extended_filter_enabled ? 1.0e-6f : o->normal_error_threshold;
    for (i = 0; i < PART_LEN1; i++) {
      float abs_ef;
      ef[0][i] /= (s->xPow[i] + 1e-10f);
      ef[1][i] /= (s->xPow[i] + 1e-10f);
      abs_ef = sqrtf(ef[0][i] * ef[0][i] + ef[1][i] * ef[1][i]);
      if (abs_ef > error_threshold) {
        abs_ef = error_threshold / (abs_ef + 1e-10f);
        ef[0][i] *= abs_ef;
        ef[1][i] *= abs_ef;
      }
      ef[0][i] *= mu;
      ef[1][i] *= mu;
    }
  }

  for (i = 0; i < o->num_partitions; i++) { /* FilterAdaptation, core:221-269 */
    int xPos = (i + s->xfBufBlockPos) * PART_LEN1;
    const int pos = i * PART_LEN1;
    int j;
    if (i + s->xfBufBlockPos >= o->num_partitions) xPos -= o->num_partitions * PART_LEN1;
    for (j = 0; j < PART_LEN; j++) {
      const float aRe = s->xfBuf[0][xPos + j], aIm = -s->xfBuf[1][xPos + j];
      fft[2 * j] = aRe * ef[0][j] - aIm * ef[1][j];
      fft[2 * j + 1] = aRe * ef[1][j] + aIm * ef[0][j];
    }
    {
      const float aRe = s->xfBuf[0][xPos + PART_LEN], aIm = -s->xfBuf[1][xPos + PART_LEN];
      fft[1] = aRe * ef[0][PART_LEN] - aIm * ef[1][PART_LEN];
    }
    asp_aec_oracle_rdft128(fft, -1);
    memset(fft + PART_LEN, 0, sizeof(float) * PART_LEN);
    {
      const float scale = 2.0f / PART_LEN2;
      for (j = 0; j < PART_LEN; j++) fft[j] *= scale;
    }
    asp_aec_oracle_rdft128(fft, 1);
    s->wfBuf[0][pos] += fft[0];
    s->wfBuf[0][pos + PART_LEN] += fft[1];
    for (j = 1; j < PART_LEN; j++) {
      s->wfBuf[0][pos + j] += fft[2 * j];
      s->wfBuf[1][pos + j] += fft[2 * j + 1];
    }
  }

  {
    float outputH[PART_LEN];
    nonlinear_processing(o, output, outputH, noisePow);
    if (o->metricsMode == 1) { /* core:1270-1275 */
      update_level(&o->met.farlevel, xf);
      update_level(&o->met.nearlevel, df);
      update_metrics(&o->met, s->echoState);
    }
    if (o->num_bands > 1) { /* core:1280-1282 */
      RingPos keep = o->out_pos;
      ring_write(&keep, o->outfrH, 1, outputH, PART_LEN);
    }
  }
  ring_write(&o->out_pos, o->outfr, 1, output, PART_LEN); /* core:1276 */
  o->blocks_processed++;
}

/* ------------------------------------------------------------ frame plumbing */
static int signal_based_delay_correction(AspAecOracle* o) { /* SignalBasedDelayCorrection, core:797-850 */
  AspAecDelayState* d = &o->de;
  int delay_correction = 0;
  const int last_delay = d->last_delay;
  if ((last_delay >= 0) && (last_delay != d->previous_delay) && (de_quality(d) > d->delay_quality_threshold)) {
    const int delay = last_delay - d->lookahead;
    if (delay <= 0 || delay > (o->num_partitions / 4)) {
      const int available_read = rp_avail_read(&o->far_pos);
      delay_correction = -(delay - d->shift_offset);
      d->shift_offset--;
      d->shift_offset = (d->shift_offset <= 1 ? 1 : d->shift_offset);
      if (delay_correction > available_read - o->mult - 1) {
        delay_correction = 0;
      } else {
        d->previous_delay = last_delay;
        ++d->delay_correction_count;
      }
    }
  }
  if (d->delay_correction_count > 0) {
    float delay_quality = de_quality(d);
    delay_quality = (delay_quality > 0.07f ? 0.07f : delay_quality); /* kDelayQualityThresholdMax */
    d->delay_quality_threshold = (delay_quality > d->delay_quality_threshold ? delay_quality : d->delay_quality_threshold);
  }
  return delay_correction;
}

static void process_frames(AspAecOracle* o, const float* nearend, const float* nearendH,
                           int num_samples, int knownDelay, float* out,
                           float* outH) { /* WebRtcAec_ProcessFrames, core:1647-1778 */
  for (int j = 0; j < num_samples; j += FRAME_LEN) {
    int out_elements;
    if (o->num_bands > 1) { /* core:1691-1693 */
      RingPos keep = o->near_pos;
      ring_write(&keep, o->nearfrH, 1, nearendH + j, FRAME_LEN);
    }
    ring_write(&o->near_pos, o->nearfr, 1, nearend + j, FRAME_LEN);
    if (o->system_delay < FRAME_LEN) far_move_read(o, -(o->mult + 1));
    if (o->reported_delay_enabled) { /* 2 a), core:1703-1718 */
      const int move_elements = (o->core_knownDelay - knownDelay - 32) / PART_LEN;
      const int moved_elements = rp_move_read(&o->far_pos, move_elements);
      o->core_knownDelay -= moved_elements * PART_LEN;
    } else { /* 2 b) signal based delay correction, core:1719-1732 */
      const int move_elements = signal_based_delay_correction(o);
      const int moved_elements = rp_move_read(&o->far_pos, move_elements);
      de_soft_reset(&o->de, moved_elements);
      /* the under-run guard of this branch, core:1747-1750 */
      if (rp_avail_read(&o->far_pos) < (o->mult + 1)) far_move_read(o, -(o->mult + 1));
    }
    while (rp_avail_read(&o->near_pos) >= PART_LEN) process_block(o);
    o->system_delay -= FRAME_LEN;
    out_elements = rp_avail_read(&o->out_pos);
    if (out_elements < FRAME_LEN) rp_move_read(&o->out_pos, out_elements - FRAME_LEN);
    if (o->num_bands > 1) { /* core:1767-1776 */
      RingPos keep = o->out_pos;
      ring_read(&keep, o->outfrH, 1, outH + j, FRAME_LEN);
    }
    ring_read(&o->out_pos, o->outfr, 1, out + j, FRAME_LEN);
  }
}

static void est_buf_delay_normal(AspAecOracle* o) { /* EstBufDelayNormal, ec:816-867 */
  const int nSampSndCard = o->msInSndCardBuf * kSampMsNb * o->rate_factor;
  int current_delay = nSampSndCard - o->system_delay;
  int delay_difference;
  current_delay += FRAME_LEN * o->rate_factor;
  if (o->skewMode == kAecTrue && o->resample == kAecTrue) current_delay -= RS_DELAY; /* ec:831-833 */
  if (current_delay < PART_LEN) current_delay += far_move_read(o, 1) * PART_LEN;
  o->filtDelay = o->filtDelay < 0 ? 0 : o->filtDelay;
  {
    const int16_t v = (int16_t)(0.8 * o->filtDelay + 0.2 * current_delay);
    o->filtDelay = v > 0 ? v : 0;
  }
  delay_difference = o->filtDelay - o->knownDelay;
  if (delay_difference > 224) {
    if (o->lastDelayDiff < 96) {
      o->timeForDelayChange = 0;
    } else {
      o->timeForDelayChange++;
    }
  } else if (delay_difference < 96 && o->knownDelay > 0) {
    if (o->lastDelayDiff > 224) {
      o->timeForDelayChange = 0;
    } else {
      o->timeForDelayChange++;
    }
  } else {
    o->timeForDelayChange = 0;
  }
  o->lastDelayDiff = (int16_t)delay_difference;
  if (o->timeForDelayChange > 25) {
    const int v = (int)o->filtDelay - 160;
    o->knownDelay = v > 0 ? v : 0;
  }
}

static void est_buf_delay_extended(AspAecOracle* o) { /* EstBufDelayExtended, ec:869-922 */
  const int reported_delay = o->msInSndCardBuf * kSampMsNb * o->rate_factor;
  int current_delay = reported_delay - o->system_delay;
  int delay_difference;
  current_delay += FRAME_LEN * o->rate_factor;
  if (o->skewMode == kAecTrue && o->resample == kAecTrue) current_delay -= RS_DELAY; /* ec:884-886 */
  if (current_delay < PART_LEN) current_delay += far_move_read(o, 2) * PART_LEN;
  if (o->filtDelay == -1) {
    const double v = 0.5 * current_delay; /* WEBRTC_SPL_MAX(0, 0.5 * current_delay) -> short */
    o->filtDelay = (int16_t)(v > 0 ? v : 0);
  } else {
    const int16_t v = (int16_t)(0.95 * o->filtDelay + 0.05 * current_delay);
    o->filtDelay = v > 0 ? v : 0;
  }
  delay_difference = o->filtDelay - o->knownDelay;
  if (delay_difference > 384) {
    if (o->lastDelayDiff < 128) {
      o->timeForDelayChange = 0;
    } else {
      o->timeForDelayChange++;
    }
  } else if (delay_difference < 128 && o->knownDelay > 0) {
    if (o->lastDelayDiff > 384) {
      o->timeForDelayChange = 0;
    } else {
      o->timeForDelayChange++;
    }
  } else {
    o->timeForDelayChange = 0;
  }
  o->lastDelayDiff = (int16_t)delay_difference;
  if (o->timeForDelayChange > 25) {
    const int v = (int)o->filtDelay - 256;
    o->knownDelay = v > 0 ? v : 0;
  }
}

/* ProcessExtended, ec:744-814 (WEBRTC_UNTRUSTED_DELAY and WEBRTC_MAC undefined: the trusted-delay branch,
 * kFixedDelayMs = 50, kMinTrustedDelayMs = 20, kDelayDiffOffsetSamples = 0, ec:70-84) */
static void process_extended(AspAecOracle* o, const float* nearend, const float* nearendH, float* out,
                             float* outH, int nrOfSamples, int16_t reported_delay_ms) {
  const int kFixedDelayMs = 50, kMinTrustedDelayMs = 20, delay_diff_offset = 0;
  reported_delay_ms = reported_delay_ms < kMinTrustedDelayMs ? kMinTrustedDelayMs : reported_delay_ms;
  reported_delay_ms = reported_delay_ms >= kMaxTrustedDelayMs ? kFixedDelayMs : reported_delay_ms;
  o->msInSndCardBuf = reported_delay_ms;
  if (!o->farend_started) {
    if (nearend != out) memcpy(out, nearend, sizeof(float) * nrOfSamples);
    if (o->num_bands > 1 && nearendH != outH) memcpy(outH, nearendH, sizeof(float) * nrOfSamples);
    return;
  }
  if (o->startup_phase) {
    const int startup_size_ms = reported_delay_ms < kFixedDelayMs ? kFixedDelayMs : reported_delay_ms;
    const int overhead_elements = (o->system_delay - startup_size_ms / 2 * o->rate_factor * 8) / PART_LEN;
    far_move_read(o, overhead_elements);
    o->startup_phase = 0;
  }
  if (o->reported_delay_enabled) est_buf_delay_extended(o); /* ec:796-798 */
  {
    const int adjusted = o->knownDelay + delay_diff_offset;
    process_frames(o, nearend, nearendH, nrOfSamples, adjusted > 0 ? adjusted : 0, out, outH);
  }
}

enum { MAX_RESAMP_LEN = 5 * FRAME_LEN }; /* ec:92 */

/* WebRtcAec_ResampleLinear, rs:74-123 */
static void resample_linear(AspAecOracle* o, const float* inspeech, int size, float skew, float* outspeech,
                            int* size_out) {
  float* y;
  float be, tnew;
  int tn, mm;
  memcpy(&o->rs_buffer[FRAME_LEN + RS_DELAY], inspeech, size * sizeof(inspeech[0]));
  be = 1 + skew;
  mm = 0;
  y = &o->rs_buffer[FRAME_LEN];
  tnew = be * mm + o->rs_position;
  tn = (int)tnew;
  while (tn < size) {
    outspeech[mm] = y[tn] + (tnew - tn) * (y[tn + 1] - y[tn]);
    mm++;
    tnew = be * mm + o->rs_position;
    tn = (int)tnew;
  }
  *size_out = mm;
  o->rs_position += (*size_out) * be - size;
  memmove(o->rs_buffer, &o->rs_buffer[size], (RS_BUF - size) * sizeof(o->rs_buffer[0]));
}

static int estimate_skew(const int* rawSkew, int size, int deviceSampleRateHz, float* skewEst) { /* rs:143-217 */
  const int absLimitOuter = (int)(0.04f * deviceSampleRateHz);
  const int absLimitInner = (int)(0.0025f * deviceSampleRateHz);
  int i = 0, n = 0, upperLimit = 0, lowerLimit = 0;
  float rawAvg = 0, err = 0, rawAbsDev = 0, cumSum = 0, x = 0, x2 = 0, y = 0, xy = 0, xAvg = 0, denom = 0, skew = 0;
  *skewEst = 0;
  for (i = 0; i < size; i++) {
    if ((rawSkew[i] < absLimitOuter && rawSkew[i] > -absLimitOuter)) {
      n++;
      rawAvg += rawSkew[i];
    }
  }
  if (n == 0) return -1;
  rawAvg /= n;
  for (i = 0; i < size; i++) {
    if ((rawSkew[i] < absLimitOuter && rawSkew[i] > -absLimitOuter)) {
      err = rawSkew[i] - rawAvg;
      rawAbsDev += err >= 0 ? err : -err;
    }
  }
  rawAbsDev /= n;
  upperLimit = (int)(rawAvg + 5 * rawAbsDev + 1);
  lowerLimit = (int)(rawAvg - 5 * rawAbsDev - 1);
  n = 0;
  for (i = 0; i < size; i++) {
    if ((rawSkew[i] < absLimitInner && rawSkew[i] > -absLimitInner) ||
        (rawSkew[i] < upperLimit && rawSkew[i] > lowerLimit)) {
      n++;
      cumSum += rawSkew[i];
      x += n;
      x2 += n * n;
      y += cumSum;
      xy += n * cumSum;
    }
  }
  if (n == 0) return -1;
  xAvg = x / n;
  denom = x2 - xAvg * x;
  if (denom != 0) skew = (xy - xAvg * y) / denom;
  *skewEst = skew;
  return 0;
}

static int get_skew(AspAecOracle* o, int rawSkew, float* skewEst) { /* WebRtcAec_GetSkew, rs:125-141 */
  int err = 0;
  if (o->rs_skewDataIndex < RS_EST) {
    o->rs_skewData[o->rs_skewDataIndex] = rawSkew;
    o->rs_skewDataIndex++;
  } else if (o->rs_skewDataIndex == RS_EST) {
    err = estimate_skew(o->rs_skewData, RS_EST, o->rs_deviceSampleRateHz, skewEst);
    o->rs_skewEstimate = *skewEst;
    o->rs_skewDataIndex++;
  } else {
    *skewEst = o->rs_skewEstimate;
  }
  return err;
}

int asp_aec_oracle_buffer_farend(AspAecOracle* o, const float* farend, int nrOfSamples) { /* ec:278-339 */
  if (farend == NULL) {
    o->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (o->initFlag != kInitCheck) {
    o->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (nrOfSamples != 80 && nrOfSamples != 160) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  {
    float new_farend[MAX_RESAMP_LEN];
    int newNrOfSamples = nrOfSamples;
    if (o->skewMode == kAecTrue && o->resample == kAecTrue) { /* ec:304-313 */
      resample_linear(o, farend, nrOfSamples, o->skew, new_farend, &newNrOfSamples);
      farend = new_farend;
    }
    o->farend_started = 1;
    o->system_delay += newNrOfSamples;
    ring_write(&o->pre_pos, o->pre, 1, farend, newNrOfSamples);
  }
  while (rp_avail_read(&o->pre_pos) >= PART_LEN2) {
    float tmp[PART_LEN2];
    ring_read(&o->pre_pos, o->pre, 1, tmp, PART_LEN2);
    buffer_farend_partition(o, tmp);
    rp_move_read(&o->pre_pos, -PART_LEN);
  }
  return 0;
}

static int process_normal(AspAecOracle* o, const float* nearend, const float* nearendH, float* out,
                          float* outH, int nrOfSamples,
                          int16_t msInSndCardBuf, int32_t skew) { /* ProcessNormal, ec:594-742 */
  const int nBlocks10ms = nrOfSamples / (FRAME_LEN * o->rate_factor);
  int retVal = 0;
  msInSndCardBuf = msInSndCardBuf > kMaxTrustedDelayMs ? kMaxTrustedDelayMs : msInSndCardBuf;
  msInSndCardBuf += 10;
  o->msInSndCardBuf = msInSndCardBuf;
  if (o->skewMode == kAecTrue) { /* ec:614-645 */
    if (o->skewFrCtr < 25) {
      o->skewFrCtr++;
    } else {
      if (get_skew(o, skew, &o->skew) == -1) {
        o->skew = 0;
        o->lastError = AEC_BAD_PARAMETER_WARNING;
        retVal = -1;
      }
      o->skew /= o->sampFactor * nrOfSamples;
      if (o->skew < 1.0e-3 && o->skew > -1.0e-3) {
        o->resample = kAecFalse;
      } else {
        o->resample = kAecTrue;
      }
      if (o->skew < -0.5f) {
        o->skew = -0.5f;
      } else if (o->skew > 1.0f) {
        o->skew = 1.0f;
      }
    }
  }
  if (o->startup_phase) {
    if (nearend != out) memcpy(out, nearend, sizeof(float) * nrOfSamples);
    if (o->num_bands > 1 && nearendH != outH) memcpy(outH, nearendH, sizeof(float) * nrOfSamples);
    if (o->checkBuffSize) {
      double lim;
      o->checkBufSizeCtr++;
      if (o->counter == 0) {
        o->firstVal = o->msInSndCardBuf;
        o->sum = 0;
      }
      lim = 0.2 * o->msInSndCardBuf;
      if (lim < kSampMsNb) lim = kSampMsNb;
      if (abs(o->firstVal - o->msInSndCardBuf) < lim) {
        o->sum += o->msInSndCardBuf;
        o->counter++;
      } else {
        o->counter = 0;
      }
      if (o->counter * nBlocks10ms >= 6) {
        const int v = (3 * o->sum * o->rate_factor * 8) / (4 * o->counter * PART_LEN);
        o->bufSizeStart = v < kMaxBufSizeStart ? v : kMaxBufSizeStart;
        o->checkBuffSize = 0;
      }
      if (o->checkBufSizeCtr * nBlocks10ms > 50) {
        const int v = (o->msInSndCardBuf * o->rate_factor * 3) / 40;
        o->bufSizeStart = v < kMaxBufSizeStart ? v : kMaxBufSizeStart;
        o->checkBuffSize = 0;
      }
    }
    if (!o->checkBuffSize) {
      const int overhead_elements = o->system_delay / PART_LEN - o->bufSizeStart;
      if (overhead_elements == 0) {
        o->startup_phase = 0;
      } else if (overhead_elements > 0) {
        far_move_read(o, overhead_elements);
        o->startup_phase = 0;
      }
    }
  } else {
    if (o->reported_delay_enabled) est_buf_delay_normal(o); /* ec:725-727 */
    process_frames(o, nearend, nearendH, nrOfSamples, o->knownDelay, out, outH);
  }
  return retVal;
}

int asp_aec_oracle_process(AspAecOracle* o, const float* nearend, float* out, int nrOfSamples,
                           int msInSndCardBuf, int32_t skew) {
  return asp_aec_oracle_process_bands(o, nearend, NULL, out, NULL, nrOfSamples, msInSndCardBuf, skew);
}

int asp_aec_oracle_process_bands(AspAecOracle* o, const float* nearend, const float* nearendH,
                                 float* out, float* outH, int nrOfSamples, int msInSndCardBuf,
                                 int32_t skew) { /* WebRtcAec_Process, ec:341-408 */
  int retVal = 0;
  if (out == NULL) {
    o->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (o->initFlag != kInitCheck) {
    o->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (nrOfSamples != 80 && nrOfSamples != 160) {
    o->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (msInSndCardBuf < 0) {
    msInSndCardBuf = 0;
    o->lastError = AEC_BAD_PARAMETER_WARNING;
    retVal = -1;
  } else if (msInSndCardBuf > kMaxTrustedDelayMs) {
    o->lastError = AEC_BAD_PARAMETER_WARNING;
    retVal = -1;
  }
  if (o->num_bands > 1 && (nearendH == NULL || outH == NULL)) {
    o->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (o->extended_filter_enabled) /* ec:377-394 */
    process_extended(o, nearend, nearendH, out, outH, nrOfSamples, (int16_t)msInSndCardBuf);
  else if (process_normal(o, nearend, nearendH, out, outH, nrOfSamples, (int16_t)msInSndCardBuf, skew) != 0)
    retVal = -1;
  return retVal;
}

int asp_aec_oracle_echo_status(const AspAecOracle* o) { return o->st.echoState; }

/* WebRtcAec_GetDelayMetrics, ec:550-571 + WebRtcAec_GetDelayMetricsCore, core:1780-1836 */
int asp_aec_oracle_get_delay_metrics(AspAecOracle* o, int* median, int* std) {
  int i, delay_values = 0, num_delay_values = 0, my_median = 0;
  const int kMsPerBlock = PART_LEN / (o->mult * 8);
  float l1_norm = 0;
  if (median == NULL || std == NULL) {
    o->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (o->initFlag != kInitCheck) {
    o->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (o->delay_logging == 0) {
    o->lastError = AEC_UNSUPPORTED_FUNCTION_ERROR;
    return -1;
  }
  for (i = 0; i < DE_HIST; i++) num_delay_values += o->de.delay_histogram[i];
  if (num_delay_values == 0) {
    *median = -1;
    *std = -1;
    return 0;
  }
  delay_values = num_delay_values >> 1;
  for (i = 0; i < DE_HIST; i++) {
    delay_values -= o->de.delay_histogram[i];
    if (delay_values < 0) {
      my_median = i;
      break;
    }
  }
  *median = (my_median - o->de.lookahead) * kMsPerBlock;
  for (i = 0; i < DE_HIST; i++) l1_norm += (float)abs(i - my_median) * o->de.delay_histogram[i];
  *std = (int)(l1_norm / (float)num_delay_values + 0.5f) * kMsPerBlock;
  memset(o->de.delay_histogram, 0, sizeof o->de.delay_histogram);
  return 0;
}

/* the delay estimator's state with the stream's far-buffer read side and system delay next to it */
void asp_aec_oracle_export_delay(const AspAecOracle* o, AspAecDelayState* d) {
  *d = o->de;
  d->far_read = o->far_pos.read;
  d->far_write = o->far_pos.write;
  d->far_wrap = o->far_pos.wrap;
  d->system_delay = o->system_delay;
}
/* the resampler's position and the skew the next BufferFarend call resamples with (rs:29-37, ec internal) */
void asp_aec_oracle_export_skew(const AspAecOracle* o, float* position, float* skew, int* resample, int* index) {
  *position = o->rs_position;
  *skew = o->skew;
  *resample = o->resample;
  *index = o->rs_skewDataIndex;
}
void asp_aec_oracle_export_metrics(const AspAecOracle* o, AspAecMetricsState* m) { *m = o->met; }

static void level_of(const AspAecStats* s, AecLevel* out) { /* ec:479-494 */
  const float kUpWeight = 0.7f;
  out->instant = (int)s->instant;
  if ((s->himean > -100) && (s->average > -100)) {
    const float dtmp = kUpWeight * s->himean + (1 - kUpWeight) * s->average;
    out->average = (int)dtmp;
  } else {
    out->average = -100;
  }
  out->max = (int)s->max;
  out->min = s->min < 100 ? (int)s->min : -100;
}

int asp_aec_oracle_get_metrics(const AspAecOracle* o, AecMetrics* metrics) { /* WebRtcAec_GetMetrics, ec:456-548 */
  int stmp;
  level_of(&o->met.erl, &metrics->erl);
  level_of(&o->met.erle, &metrics->erle);
  stmp = (metrics->erl.average > -100 && metrics->erle.average > -100)
             ? metrics->erl.average + metrics->erle.average
             : -100;
  metrics->rerl.average = stmp;
  metrics->rerl.instant = stmp;
  metrics->rerl.max = stmp;
  metrics->rerl.min = stmp;
  level_of(&o->met.aNlp, &metrics->aNlp);
  return 0;
}
int asp_aec_oracle_error_code(const AspAecOracle* o) { return o->lastError; }

void asp_aec_oracle_export(const AspAecOracle* o, AspAecState* st, AspAecControl* c) {
  if (st) *st = o->st;
  if (c) {
    c->startup_phase = o->startup_phase;
    c->checkBuffSize = o->checkBuffSize;
    c->bufSizeStart = o->bufSizeStart;
    c->knownDelay = o->knownDelay;
    c->filtDelay = o->filtDelay;
    c->timeForDelayChange = o->timeForDelayChange;
    c->lastDelayDiff = o->lastDelayDiff;
    c->counter = o->counter;
    c->sum = o->sum;
    c->firstVal = o->firstVal;
    c->checkBufSizeCtr = o->checkBufSizeCtr;
    c->system_delay = o->system_delay;
    c->core_knownDelay = o->core_knownDelay;
    c->far_read = o->far_pos.read;
    c->far_write = o->far_pos.write;
    c->far_wrap = o->far_pos.wrap;
    c->pre_read = o->pre_pos.read;
    c->pre_write = o->pre_pos.write;
    c->pre_wrap = o->pre_pos.wrap;
    c->near_read = o->near_pos.read;
    c->near_write = o->near_pos.write;
    c->near_wrap = o->near_pos.wrap;
    c->out_read = o->out_pos.read;
    c->out_write = o->out_pos.write;
    c->out_wrap = o->out_pos.wrap;
    c->blocks_processed = o->blocks_processed;
  }
}

void asp_aec_oracle_import(AspAecOracle* o, const AspAecState* st) { o->st = *st; }

int asp_aec_oracle_run(AspAecOracle* o, const float* far, const float* near, float* out, int F,
                       int n, int delay_ms) {
  int rc = 0;
  for (int f = 0; f < F; ++f) {
    rc |= asp_aec_oracle_buffer_farend(o, far + (size_t)f * n, n);
    rc |= asp_aec_oracle_process(o, near + (size_t)f * n, out + (size_t)f * n, n, delay_ms, 0);
  }
  return rc;
}

/* ---------------------------------------------------------------- threaded */
typedef struct Shard {
  int s0, s1, S, F, n, delay_ms, rc;
  int32_t fs;
  const float *far, *near;
  float* out;
} Shard;

static void* shard_main(void* p) {
  Shard* sh = (Shard*)p;
  for (int s = sh->s0; s < sh->s1; ++s) {
    AspAecOracle* o = asp_aec_oracle_create();
    if (!o || asp_aec_oracle_init(o, sh->fs, 48000) != 0) {
      sh->rc = -1;
      asp_aec_oracle_free(o);
      return NULL;
    }
    for (int f = 0; f < sh->F; ++f) {
      const size_t off = ((size_t)f * sh->S + s) * sh->n;
      asp_aec_oracle_buffer_farend(o, sh->far + off, sh->n);
      asp_aec_oracle_process(o, sh->near + off, sh->out + off, sh->n, sh->delay_ms, 0);
    }
    asp_aec_oracle_free(o);
  }
  return NULL;
}

int asp_aec_oracle_run_mt(int num_streams, const float* far, const float* near, float* out, int F,
                          int n, int delay_ms, int32_t fs, int threads) {
  pthread_t tid[256];
  Shard sh[256];
  int rc = 0;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if (threads > num_streams) threads = num_streams;
  for (int t = 0; t < threads; ++t) {
    sh[t].s0 = (int)((long long)num_streams * t / threads);
    sh[t].s1 = (int)((long long)num_streams * (t + 1) / threads);
    sh[t].S = num_streams;
    sh[t].F = F;
    sh[t].n = n;
    sh[t].delay_ms = delay_ms;
    sh[t].fs = fs;
    sh[t].far = far;
    sh[t].near = near;
    sh[t].out = out;
    sh[t].rc = 0;
    pthread_create(&tid[t], NULL, shard_main, &sh[t]);
  }
  for (int t = 0; t < threads; ++t) {
    pthread_join(tid[t], NULL);
    rc |= sh[t].rc;
  }
  return rc;
}
