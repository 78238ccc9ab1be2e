/*
 * ns_oracle.c -- CPU restatement of the reference's float noise suppressor
 * (WebRTC NS: the 160 / 256 / 129 geometry of 16, 32 and 48 kHz and the 80 / 128 / 65 geometry of
 * 8 kHz, ns_core.c:89-98).  TEST INFRASTRUCTURE ONLY -- see ns_oracle.h.
 *
 * Parity status: PINNED against the reference C compiled from /root/reference
 * (oracle/Makefile -> oracle/_ref/libns_ref.so) and the committed golden
 * vectors tests/golden/ns_*.npz.  In ASP_NS_REDUCE_SEQ mode every float
 * operation is performed in the reference's order, so outputs and state are
 * bit-identical to the reference build with -ffp-contract=off.
 *
 * All citations are relative to
 *   /root/reference/WebRtc_AMP_Port/webrtc/modules/audio_processing/
 */
#include "ns_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXBINS ASP_NS_BINS
#define MAXANAL ASP_NS_ANAL
#define MAXBLOCKL ASP_NS_BLOCKL
#define SIMULT ASP_NS_SIMULT
/* the geometry of a stream (ns_core.c:89-98): block / analysis window / bins, as locals of that name */
#define GEO_FS(fs_)                                       \
  const int BLOCKL = (fs_) == 8000 ? 80 : 160;            \
  const int ANAL = (fs_) == 8000 ? 128 : 256;             \
  const int BINS = ANAL / 2 + 1;                          \
  (void)BLOCKL;                                           \
  (void)BINS
#define GEO(s_) GEO_FS((s_)->fs)
#define HIST ASP_NS_HIST

/* ns/defines.h:19-48 -- written with the same (float)<double literal> casts */
#define QUANTILE (float)0.25
#define END_STARTUP_LONG 200
#define END_STARTUP_SHORT 50
#define FACTOR (float)40.0
#define WIDTH (float)0.01
#define DD_PR_SNR (float)0.98
#define LRT_TAVG (float)0.50
#define SPECT_FL_TAVG (float)0.30
#define SPECT_DIFF_TAVG (float)0.30
#define PRIOR_UPDATE (float)0.10
#define NOISE_UPDATE (float)0.90
#define SPEECH_UPDATE (float)0.99
#define WIDTH_PR_MAP (float)4.0
#define LRT_FEATURE_THR (float)0.5
#define SF_FEATURE_THR (float)0.5
#define PROB_RANGE (float)0.20
#define GAMMA_PAUSE (float)0.05
#define B_LIM (float)0.5
#define K_START_BAND 5 /* ns_core.c:1045 */

/* ------------------------------------------------------------------ tables */

static float g_window[MAXANAL]; /* kBlocks160w256, ns/windows_private.h:94-147 */
static float g_window8[128];    /* kBlocks80w128, ns/windows_private.h:64-91    */
static float g_w[64];           /* makewt(64), utility/fft4g.c:642-669          */
static float g_c[64];           /* makect(64), utility/fft4g.c:671-690          */
static float g_w8[32];          /* makewt(32): the tables of WebRtc_rdft(128)   */
static float g_c8[32];          /* makect(32)                                   */
static float g_logi[MAXBINS];   /* (float)log((float)i), ns_core.c:1093         */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static unsigned bitrev(unsigned x, int bits) {
  unsigned r = 0;
  for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1u) << (bits - 1 - b);
  return r;
}

/* The reference table is sin(pi*i/192) over the 96-sample ramps printed with
 * 8 decimals and re-read as (float)<double literal>; reproduce that text
 * round trip (checked value-for-value against the header text in
 * tests/test_ns_tables.py). */
static float window_entry(int i) {
  char buf[32];
  double s;
  if (i >= 96 && i <= 160) return 1.0f;
  s = sin(M_PI * (double)(i < 96 ? i : 256 - i) / 192.0);
  snprintf(buf, sizeof buf, "%.8f", s);
  return (float)strtod(buf, NULL);
}
/* kBlocks80w128: sin(pi*i/96) over the 48-sample ramps, the same text round trip (every entry equals the
 * header's, checked when this was written and by tests/test_ns_oracle.py against the compiled reference) */
static float window8_entry(int i) {
  char buf[32];
  double s;
  if (i >= 48 && i <= 80) return 1.0f;
  s = sin(M_PI * (double)(i < 48 ? i : 128 - i) / 96.0);
  snprintf(buf, sizeof buf, "%.8f", s);
  return (float)strtod(buf, NULL);
}
/* makewt(nw) + its bitrv2, and makect(nc) (fft4g.c:642-690) for nw = nc = N: N/2 complex entries */
static void make_wt_ct(int N, float* w, float* c) {
  float tmp[64];
  const int nwh = N >> 1;
  int j, bits = 0;
  float delta = (float)atan(1.0f) / nwh;
  while ((1 << bits) < nwh) ++bits;
  tmp[0] = 1;
  tmp[1] = 0;
  tmp[nwh] = (float)cos(delta * nwh);
  tmp[nwh + 1] = tmp[nwh];
  for (j = 2; j < nwh; j += 2) {
    float x = (float)cos(delta * j);
    float y = (float)sin(delta * j);
    tmp[j] = x;
    tmp[j + 1] = y;
    tmp[N - j] = y;
    tmp[N - j + 1] = x;
  }
  for (j = 0; j < nwh; ++j) {
    unsigned r = bitrev((unsigned)j, bits);
    w[2 * j] = tmp[2 * r];
    w[2 * j + 1] = tmp[2 * r + 1];
  }
  c[0] = (float)cos(delta * nwh);
  c[nwh] = 0.5f * c[0];
  for (j = 1; j < nwh; j++) {
    c[j] = 0.5f * (float)cos(delta * j);
    c[N - j] = 0.5f * (float)sin(delta * j);
  }
}

static void build_tables(void) {
  int j;
  float tmp[64];
  /* makewt(nw = 64): fft4g.c:647-666 */
  {
    const int nw = 64, nwh = 32;
    float delta = (float)atan(1.0f) / nwh;
    tmp[0] = 1;
    tmp[1] = 0;
    tmp[nwh] = (float)cos(delta * nwh);
    tmp[nwh + 1] = tmp[nwh];
    for (j = 2; j < nwh; j += 2) {
      float x = (float)cos(delta * j);
      float y = (float)sin(delta * j);
      tmp[j] = x;
      tmp[j + 1] = y;
      tmp[nw - j] = y;
      tmp[nw - j + 1] = x;
    }
    /* bitrv2(nw, ip + 2, w) is the bit reversal of the 32 complex entries */
    for (j = 0; j < 32; ++j) {
      unsigned r = bitrev((unsigned)j, 5);
      g_w[2 * j] = tmp[2 * r];
      g_w[2 * j + 1] = tmp[2 * r + 1];
    }
  }
  /* makect(nc = 64): fft4g.c:676-686 */
  {
    const int nc = 64, nch = 32;
    float delta = (float)atan(1.0f) / nch;
    g_c[0] = (float)cos(delta * nch);
    g_c[nch] = 0.5f * g_c[0];
    for (j = 1; j < nch; j++) {
      g_c[j] = 0.5f * (float)cos(delta * j);
      g_c[nc - j] = 0.5f * (float)sin(delta * j);
    }
  }
  for (j = 0; j < MAXANAL; ++j) g_window[j] = window_entry(j);
  for (j = 0; j < 128; ++j) g_window8[j] = window8_entry(j);
  make_wt_ct(32, g_w8, g_c8);
  g_logi[0] = 0.f;
  for (j = 1; j < MAXBINS; ++j) g_logi[j] = (float)log((float)j);
}

static void ensure_tables(void) { pthread_once(&g_once, build_tables); }

const float* asp_ns_oracle_window(void) { ensure_tables(); return g_window; }
const float* asp_ns_oracle_fft_w(void) { ensure_tables(); return g_w; }
const float* asp_ns_oracle_fft_c(void) { ensure_tables(); return g_c; }
const float* asp_ns_oracle_window8(void) { ensure_tables(); return g_window8; }
const float* asp_ns_oracle_fft_w8(void) { ensure_tables(); return g_w8; }
const float* asp_ns_oracle_fft_c8(void) { ensure_tables(); return g_c8; }

/* --------------------------------------------------------------------- FFT */

/* One radix-4 pass over N complex points at complex stride l: cft1st (l = 1,
 * fft4g.c:1002-1104) and cftmdl (l = 4, 16, ...; fft4g.c:1107-1231) are the
 * same pass; blocks of 4l points are numbered B = 0.. and
 *   B = 0      no twiddles                 (:1008-1023, :1114-1134)
 *   B = 1      the w[2] = cos(pi/4) block  (:1024-1044, :1135-1160)
 *   B = 2u     twiddles w[2u], w[4u]       (:1049-1076, :1166-1198)
 *   B = 2u+1   twiddles w[2u], w[4u+2]     (:1077-1102, :1199-1229)      */
static void cft_pass(float* a, int N, int l, const float* w) {
  const int bs = 4 * l;
  for (int B = 0; B * bs < N; ++B) {
    int kind = 2;
    float ws = 0.f, w1r = 0.f, w1i = 0.f, w2r = 0.f, w2i = 0.f, w3r = 0.f, w3i = 0.f;
    if (B == 0) {
      kind = 0;
    } else if (B == 1) {
      kind = 1;
      ws = w[2];
    } else {
      const int u = B >> 1;
      const float wk2r = w[2 * u], wk2i = w[2 * u + 1];
      if ((B & 1) == 0) {
        w1r = w[4 * u];
        w1i = w[4 * u + 1];
        w3r = w1r - 2 * wk2i * w1i;
        w3i = 2 * wk2i * w1r - w1i;
        w2r = wk2r;
        w2i = wk2i;
      } else {
        w1r = w[4 * u + 2];
        w1i = w[4 * u + 3];
        w3r = w1r - 2 * wk2r * w1i;
        w3i = 2 * wk2r * w1r - w1i;
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

/* bitrv2 (fft4g.c:693-790) is the bit reversal of the complex index. */
static void bit_reverse(float* a, int N, int bits) {
  for (int i = 0; i < N; ++i) {
    int r = (int)bitrev((unsigned)i, bits);
    if (r > i) {
      float tr = a[2 * i], ti = a[2 * i + 1];
      a[2 * i] = a[2 * r];
      a[2 * i + 1] = a[2 * r + 1];
      a[2 * r] = tr;
      a[2 * r + 1] = ti;
    }
  }
}

/* cftfsub / cftbsub (fft4g.c:902-949, 952-999) for N complex points (n = 2N floats), N = 128 or 64:
 * radix-4 passes at complex strides 1, 4, .. while 4 l < N, then
 *   N = 128 (n = 256): the radix-2 tail at stride 64 (:939-947 / :989-997);
 *   N = 64  (n = 128): one twiddle-free radix-4 stage at stride 16 (:918-937 / :968-987; forward it is
 *                      the B = 0 case of the general pass, backward it carries the conjugation). */
static void cftN(float* a, int N, const float* w, int backward) {
  int l = 1;
  cft_pass(a, N, l, w);
  for (l = 4; 4 * l < N; l *= 4) cft_pass(a, N, l, w);
  if (4 * l == N) {
    if (!backward) {
      cft_pass(a, N, l, w); /* a single block, B = 0: no twiddles */
    } else {
      for (int q = 0; q < l; ++q) { /* :968-987 */
        float* e0 = a + 2 * q;
        float* e1 = e0 + 2 * l;
        float* e2 = e1 + 2 * l;
        float* e3 = e2 + 2 * l;
        float x0r = e0[0] + e1[0], x0i = -e0[1] - e1[1];
        float x1r = e0[0] - e1[0], x1i = -e0[1] + e1[1];
        float x2r = e2[0] + e3[0], x2i = e2[1] + e3[1];
        float x3r = e2[0] - e3[0], x3i = e2[1] - e3[1];
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
    return;
  }
  for (int q = 0; q < l; ++q) { /* 2 l == N */
    float* lo = a + 2 * q;
    float* hi = a + 2 * (q + l);
    if (!backward) { /* :939-947 */
      float x0r = lo[0] - hi[0];
      float x0i = lo[1] - hi[1];
      lo[0] += hi[0];
      lo[1] += hi[1];
      hi[0] = x0r;
      hi[1] = x0i;
    } else { /* :989-997 */
      float x0r = lo[0] - hi[0];
      float x0i = -lo[1] + hi[1];
      lo[0] += hi[0];
      lo[1] = -lo[1] - hi[1];
      hi[0] = x0r;
      hi[1] = x0i;
    }
  }
}

/* WebRtc_rdft(n, isgn, a, ip, w) for n = 256 or 128 (fft4g.c:324-362; ks = 1 in rftfsub / rftbsub) */
static void rdft_n(float* a, int n, int isgn) {
  const int m = n >> 1, nc = n >> 2, bits = n == 256 ? 7 : 6;
  const float* w = n == 256 ? g_w : g_w8;
  const float* c = n == 256 ? g_c : g_c8;
  ensure_tables();
  if (isgn >= 0) { /* fft4g.c:339-349 */
    float xi;
    bit_reverse(a, m, bits);
    cftN(a, m, w, 0);
    for (int p = 1; p < nc; ++p) { /* rftfsub, fft4g.c:1234-1256 */
      const int j = 2 * p, k = n - j;
      float wkr = 0.5f - c[nc - p];
      float wki = c[p];
      float xr = a[j] - a[k];
      float xim = a[j + 1] + a[k + 1];
      float yr = wkr * xr - wki * xim;
      float yi = wkr * xim + wki * xr;
      a[j] -= yr;
      a[j + 1] -= yi;
      a[k] += yr;
      a[k + 1] -= yi;
    }
    xi = a[0] - a[1];
    a[0] += a[1];
    a[1] = xi;
  } else { /* fft4g.c:350-360 */
    a[1] = 0.5f * (a[0] - a[1]);
    a[0] -= a[1];
    a[1] = -a[1]; /* rftbsub, fft4g.c:1259-1283 */
    for (int p = 1; p < nc; ++p) {
      const int j = 2 * p, k = n - j;
      float wkr = 0.5f - c[nc - p];
      float wki = c[p];
      float xr = a[j] - a[k];
      float xim = a[j + 1] + a[k + 1];
      float yr = wkr * xr + wki * xim;
      float yi = wkr * xim - wki * xr;
      a[j] -= yr;
      a[j + 1] = yi - a[j + 1];
      a[k] += yr;
      a[k + 1] = yi - a[k + 1];
    }
    a[m + 1] = -a[m + 1];
    bit_reverse(a, m, bits);
    cftN(a, m, w, 1);
  }
}

void asp_ns_oracle_rdft256(float* a, int isgn) { rdft_n(a, 256, isgn); }
void asp_ns_oracle_rdft128(float* a, int isgn) { rdft_n(a, 128, isgn); }

/* -------------------------------------------------------------- reductions */

/* The wave64 association used by the HIP kernels: 64 partials, butterfly
 * xor 1,2,4,8,16,32 (every lane ends with the same value). */
static float butterfly64(float* t) {
  for (int m = 1; m <= 32; m <<= 1) {
    float u[64];
    for (int l = 0; l < 64; ++l) u[l] = t[l] + t[l ^ m];
    memcpy(t, u, sizeof u);
  }
  return t[0];
}

/* The 32-lane association of the two-streams-per-wave kernel (ns_kernels2.hip): 32 partials,
 * butterfly xor 1,2,4,8,16. */
static float butterfly32(float* t) {
  for (int m = 1; m <= 16; m <<= 1) {
    float u[32];
    for (int l = 0; l < 32; ++l) u[l] = t[l] + t[l ^ m];
    memcpy(t, u, sizeof u);
  }
  return t[0];
}

/* Sum of x[0..BINS-1] (one value per bin).  TREE: lane l holds bins l and l+64,
 * lane 0 additionally bin 128.  TREE32: lane l < 16 holds bins l, l+16, l+32, l+48, lane 16+l holds
 * 64+l, 80+l, 96+l, 112+l (summed in that order); bin 128 is added to the butterfly's result.  TREE64P: see below.  With 65 bins (8 kHz) every device association is
 * the one of ns_kernels.hip's 8 kHz instantiation: lane l holds bin l, lane 0 additionally bin 64. */
static float sum_bins(const float* x, int mode, int BINS) {
  if (BINS == 65 && mode != ASP_NS_REDUCE_SEQ) {
    float t[64];
    for (int l = 0; l < 64; ++l) t[l] = x[l];
    t[0] = t[0] + x[64];
    return butterfly64(t);
  }
  if (mode == ASP_NS_REDUCE_TREE32) {
    float t[32];
    for (int l = 0; l < 32; ++l) {
      const int b = (l & 15) + 64 * (l >> 4);
      t[l] = ((x[b] + x[b + 16]) + x[b + 32]) + x[b + 48];
    }
    return butterfly32(t) + x[128];
  }
  if (mode == ASP_NS_REDUCE_TREE64P) {
    /* ns_kernels1.hip: lane l = 2 lam + h holds bins q + 64 g + 16 h and that + 32
     * (lam = q + 16 g); bin 128 is added to the butterfly's result */
    float t[64];
    for (int l = 0; l < 64; ++l) {
      const int lam = l >> 1, h = l & 1;
      const int b = (lam & 15) + 64 * (lam >> 4) + 16 * h;
      t[l] = x[b] + x[b + 32];
    }
    return butterfly64(t) + x[128];
  }
  if (mode == ASP_NS_REDUCE_SEQ) {
    float s = 0.f;
    for (int i = 0; i < BINS; ++i) s += x[i];
    return s;
  } else {
    float t[64];
    for (int l = 0; l < 64; ++l) t[l] = x[l] + x[l + 64];
    t[0] = t[0] + x[128];
    return butterfly64(t);
  }
}

/* Energy of an ANAL-sample buffer (ns_core.c:951-960).  TREE: `by4` selects the
 * lane layout: 1 = lane l holds samples 4l..4l+3 (analysis side), 0 = lane l
 * holds samples 2l, 2l+1, 2l+128, 2l+129 (after the inverse FFT).  128 samples (8 kHz): lane l holds
 * samples 2l, 2l+1 on both sides. */
static float energy256(const float* x, int mode, int by4, int ANAL) {
  if (ANAL == 128 && mode != ASP_NS_REDUCE_SEQ) {
    float t[64];
    for (int l = 0; l < 64; ++l) {
      float s = x[2 * l] * x[2 * l];
      s += x[2 * l + 1] * x[2 * l + 1];
      t[l] = s;
    }
    return butterfly64(t);
  }
  if (mode == ASP_NS_REDUCE_TREE32) {
    /* analysis side: lane l holds samples 8l..8l+7; synthesis side: lane l holds complex
     * elements p = (l & 15) + 16 t + 64 (l >> 4), t = 0..3, i.e. samples 2p, 2p+1 */
    float t[32];
    for (int l = 0; l < 32; ++l) {
      float s = 0.f;
      for (int k = 0; k < 8; ++k) {
        const int idx = by4 ? 8 * l + k : 2 * ((l & 15) + 16 * (k >> 1) + 64 * (l >> 4)) + (k & 1);
        s = k == 0 ? x[idx] * x[idx] : s + x[idx] * x[idx];
      }
      t[l] = s;
    }
    return butterfly32(t);
  }
  if (mode == ASP_NS_REDUCE_TREE64P && !by4) {
    /* synthesis side of ns_kernels1.hip: lane l = 2 lam + h holds complex elements
     * E = q + 64 g + 16 h and E + 32, i.e. samples 2E, 2E+1, 2E+64, 2E+65 (the analysis side
     * is the TREE layout: samples 4l .. 4l+3) */
    float t[64];
    for (int l = 0; l < 64; ++l) {
      const int lam = l >> 1, h = l & 1;
      const float* p = x + 2 * ((lam & 15) + 64 * (lam >> 4) + 16 * h);
      float s = p[0] * p[0];
      s += p[1] * p[1];
      s += p[64] * p[64];
      s += p[65] * p[65];
      t[l] = s;
    }
    return butterfly64(t);
  }
  if (mode == ASP_NS_REDUCE_SEQ) {
    float e = 0.f;
    for (int i = 0; i < ANAL; ++i) e += x[i] * x[i];
    return e;
  } else {
    float t[64];
    for (int l = 0; l < 64; ++l) {
      const float* p = by4 ? x + 4 * l : x + 2 * l;
      const int o2 = by4 ? 2 : 128, o3 = by4 ? 3 : 129;
      float s = p[0] * p[0];
      s += p[1] * p[1];
      s += p[o2] * p[o2];
      s += p[o3] * p[o3];
      t[l] = s;
    }
    return butterfly64(t);
  }
}

/* ------------------------------------------------------------- init/policy */

int asp_ns_oracle_set_policy(AspNsState* s, int mode) { /* ns_core.c:1013-1041 */
  if (s == NULL || mode < 0 || mode > 3) return -1;
  s->aggrMode = mode;
  if (mode == 0) {
    s->overdrive = 1.f;
    s->denoiseBound = 0.5f;
    s->gainmap = 0;
  } else if (mode == 1) {
    s->overdrive = 1.f;
    s->denoiseBound = 0.25f;
    s->gainmap = 1;
  } else if (mode == 2) {
    s->overdrive = 1.1f;
    s->denoiseBound = 0.125f;
    s->gainmap = 1;
  } else {
    s->overdrive = 1.25f;
    s->denoiseBound = 0.09f;
    s->gainmap = 1;
  }
  return 0;
}

int asp_ns_oracle_init(AspNsState* s, uint32_t fs) { /* ns_core.c:74-214 */
  int i;
  if (s == NULL) return -1;
  if (fs != 8000 && fs != 16000 && fs != 32000 && fs != 48000) return -1; /* :82-86 */
  GEO_FS(fs);
  ensure_tables();
  memset(s, 0, sizeof *s);
  s->fs = (int32_t)fs;
  for (i = 0; i < SIMULT * MAXBINS; i++) { /* :116-119: the whole arrays, whatever magnLen is */
    s->lquantile[i] = 8.f;
    s->density[i] = 0.3f;
  }
  for (i = 0; i < SIMULT; i++) /* :121-124 */
    s->counter[i] = (int)floor((float)(END_STARTUP_LONG * (i + 1)) / (float)SIMULT);
  s->updates = 0;
  for (i = 0; i < MAXBINS; i++) s->smooth[i] = 1.f;       /* :129-131 */
  s->priorSpeechProb = 0.5f;                               /* :137 */
  for (i = 0; i < MAXBINS; i++) s->logLrtTimeAvg[i] = LRT_FEATURE_THR; /* :152-155 */
  s->featureData[0] = SF_FEATURE_THR;                      /* :159-168 */
  s->featureData[3] = LRT_FEATURE_THR;
  s->featureData[4] = SF_FEATURE_THR;
  s->blockInd = -1;                                        /* :176 */
  s->priorModelPars[0] = LRT_FEATURE_THR;                  /* :178-190 */
  s->priorModelPars[1] = 0.5f;
  s->priorModelPars[2] = 1.f;
  s->priorModelPars[3] = 0.5f;
  s->priorModelPars[4] = 1.f;
  s->priorModelPars[5] = 0.f;
  s->priorModelPars[6] = 0.f;
  s->modelUpdatePars[0] = 2;                               /* :194-199 */
  s->modelUpdatePars[1] = 500;
  s->modelUpdatePars[2] = 0;
  s->modelUpdatePars[3] = s->modelUpdatePars[1];
  asp_ns_oracle_set_policy(s, 0);                          /* :210 */
  s->initFlag = 1;
  return 0;
}

/* ------------------------------------------------- feature-extraction pars */

/* set_feature_extraction_parameters, ns_core.c:23-71 (constants only). */
#define BIN_SIZE_LRT 0.1f
#define BIN_SIZE_SPEC_FLAT 0.05f
#define BIN_SIZE_SPEC_DIFF 0.1f
#define RANGE_AVG_HIST_LRT 1.f
#define FACTOR1_MODEL_PARS 1.2f
#define FACTOR2_MODEL_PARS 0.9f
#define THRES_POS_SPEC_FLAT 0.6f
#define LIMIT_PEAK_WEIGHTS 0.5f
#define THRES_FLUCT_LRT 0.05f
#define MAX_LRT 1.f
#define MIN_LRT 0.2f
#define MAX_SPEC_FLAT 0.95f
#define MIN_SPEC_FLAT 0.1f
#define MAX_SPEC_DIFF 1.f
#define MIN_SPEC_DIFF 0.16f

/* Two dominant peaks of a 1000-bin histogram, scan order semantics of
 * ns_core.c:386-404 / :414-432. */
static void two_peaks(const int32_t* hist, float binSize, float* pos1,
                      float* pos2, int* wt1, int* wt2) {
  int maxPeak1 = 0, maxPeak2 = 0;
  *pos1 = 0.f;
  *pos2 = 0.f;
  *wt1 = 0;
  *wt2 = 0;
  for (int i = 0; i < HIST; i++) {
    float binMid = ((float)i + 0.5f) * binSize;
    if (hist[i] > maxPeak1) {
      maxPeak2 = maxPeak1;
      *wt2 = *wt1;
      *pos2 = *pos1;
      maxPeak1 = hist[i];
      *wt1 = hist[i];
      *pos1 = binMid;
    } else if (hist[i] > maxPeak2) {
      maxPeak2 = hist[i];
      *wt2 = hist[i];
      *pos2 = binMid;
    }
  }
}

/* FeatureParameterExtraction(self, 1), ns_core.c:337-517. */
static void extract_parameters(AspNsState* s) {
  int i, useFlat, useDiff, numHistLrt = 0;
  float avgHistLrt = 0.f, avgHistLrtCompl = 0.f, avgSquareHistLrt = 0.f, fluctLrt;
  float pos1F, pos2F, pos1D, pos2D, featureSum;
  int w1F, w2F, w1D, w2D;
  const float limitSpacingFlat = 2 * BIN_SIZE_SPEC_FLAT; /* :44-47 */
  const float limitSpacingDiff = 2 * BIN_SIZE_SPEC_DIFF;
  const int thresWeightFlat = (int)(0.3 * (s->modelUpdatePars[1])); /* :67-70 */
  const int thresWeightDiff = (int)(0.3 * (s->modelUpdatePars[1]));

  for (i = 0; i < HIST; i++) { /* :344-352 */
    float binMid = ((float)i + 0.5f) * BIN_SIZE_LRT;
    if (binMid <= RANGE_AVG_HIST_LRT) {
      avgHistLrt += s->histLrt[i] * binMid;
      numHistLrt += s->histLrt[i];
    }
    avgSquareHistLrt += s->histLrt[i] * binMid * binMid;
    avgHistLrtCompl += s->histLrt[i] * binMid;
  }
  if (numHistLrt > 0) avgHistLrt = avgHistLrt / ((float)numHistLrt);
  avgHistLrtCompl = avgHistLrtCompl / ((float)s->modelUpdatePars[1]);
  avgSquareHistLrt = avgSquareHistLrt / ((float)s->modelUpdatePars[1]);
  fluctLrt = avgSquareHistLrt - avgHistLrt * avgHistLrtCompl;
  if (fluctLrt < THRES_FLUCT_LRT) { /* :360-373 */
    s->priorModelPars[0] = MAX_LRT;
  } else {
    s->priorModelPars[0] = FACTOR1_MODEL_PARS * avgHistLrt;
    if (s->priorModelPars[0] < MIN_LRT) s->priorModelPars[0] = MIN_LRT;
    if (s->priorModelPars[0] > MAX_LRT) s->priorModelPars[0] = MAX_LRT;
  }

  two_peaks(s->histSpecFlat, BIN_SIZE_SPEC_FLAT, &pos1F, &pos2F, &w1F, &w2F);
  two_peaks(s->histSpecDiff, BIN_SIZE_SPEC_DIFF, &pos1D, &pos2D, &w1D, &w2D);

  useFlat = 1; /* :435-463 */
  if ((fabs(pos2F - pos1F) < limitSpacingFlat) && (w2F > LIMIT_PEAK_WEIGHTS * w1F)) {
    w1F += w2F;
    pos1F = 0.5f * (pos1F + pos2F);
  }
  if (w1F < thresWeightFlat || pos1F < THRES_POS_SPEC_FLAT) useFlat = 0;
  if (useFlat == 1) {
    s->priorModelPars[1] = FACTOR2_MODEL_PARS * pos1F;
    if (s->priorModelPars[1] < MIN_SPEC_FLAT) s->priorModelPars[1] = MIN_SPEC_FLAT;
    if (s->priorModelPars[1] > MAX_SPEC_FLAT) s->priorModelPars[1] = MAX_SPEC_FLAT;
  }

  useDiff = 1; /* :467-498 */
  if ((fabs(pos2D - pos1D) < limitSpacingDiff) && (w2D > LIMIT_PEAK_WEIGHTS * w1D)) {
    w1D += w2D;
    pos1D = 0.5f * (pos1D + pos2D);
  }
  s->priorModelPars[3] = FACTOR1_MODEL_PARS * pos1D;
  if (w1D < thresWeightDiff) useDiff = 0;
  if (s->priorModelPars[3] < MIN_SPEC_DIFF) s->priorModelPars[3] = MIN_SPEC_DIFF;
  if (s->priorModelPars[3] > MAX_SPEC_DIFF) s->priorModelPars[3] = MAX_SPEC_DIFF;
  if (fluctLrt < THRES_FLUCT_LRT) useDiff = 0;

  featureSum = (float)(1 + useFlat + useDiff); /* :504-507 */
  s->priorModelPars[4] = 1.f / featureSum;
  s->priorModelPars[5] = ((float)useFlat) / featureSum;
  s->priorModelPars[6] = ((float)useDiff) / featureSum;

  if (s->modelUpdatePars[0] >= 1) { /* :510-516 */
    memset(s->histLrt, 0, sizeof s->histLrt);
    memset(s->histSpecFlat, 0, sizeof s->histSpecFlat);
    memset(s->histSpecDiff, 0, sizeof s->histSpecDiff);
  }
}

/* FeatureParameterExtraction(self, 0), ns_core.c:309-334. */
static void update_histograms(AspNsState* s) {
  int i;
  if ((s->featureData[3] < HIST * BIN_SIZE_LRT) && (s->featureData[3] >= 0.0)) {
    i = (int)(s->featureData[3] / BIN_SIZE_LRT);
    s->histLrt[i]++;
  }
  if ((s->featureData[0] < HIST * BIN_SIZE_SPEC_FLAT) && (s->featureData[0] >= 0.0)) {
    i = (int)(s->featureData[0] / BIN_SIZE_SPEC_FLAT);
    s->histSpecFlat[i]++;
  }
  if ((s->featureData[4] < HIST * BIN_SIZE_SPEC_DIFF) && (s->featureData[4] >= 0.0)) {
    i = (int)(s->featureData[4] / BIN_SIZE_SPEC_DIFF);
    s->histSpecDiff[i]++;
  }
}

/* ---------------------------------------------------------------- analysis */

/* UpdateBuffer + Windowing (ns_core.c:855-873, 969-978). */
static void slide_and_window(float* buf, const float* frame, float* win, int fs) {
  GEO_FS(fs);
  const float* g_win = fs == 8000 ? g_window8 : g_window;
  memmove(buf, buf + BLOCKL, sizeof(float) * (ANAL - BLOCKL));
  if (frame)
    memcpy(buf + ANAL - BLOCKL, frame, sizeof(float) * BLOCKL);
  else
    memset(buf + ANAL - BLOCKL, 0, sizeof(float) * BLOCKL);
  if (win)
    for (int i = 0; i < ANAL; ++i) win[i] = g_win[i] * buf[i];
}

/* FFT(), ns_core.c:886-911. */
static void forward_spectrum(float* td, float* re, float* im, float* magn, int fs) {
  GEO_FS(fs);
  rdft_n(td, ANAL, 1);
  im[0] = 0;
  re[0] = td[0];
  magn[0] = (float)(fabs(re[0]) + 1.f);
  im[BINS - 1] = 0;
  re[BINS - 1] = td[1];
  magn[BINS - 1] = (float)(fabs(re[BINS - 1]) + 1.f);
  for (int i = 1; i < BINS - 1; ++i) {
    re[i] = td[2 * i];
    im[i] = td[2 * i + 1];
    magn[i] = sqrtf(re[i] * re[i] + im[i] * im[i]) + 1.f;
  }
}

void asp_ns_oracle_analyze(AspNsState* s, const float* frame, int mode) {
  GEO(s);
  int i, k, offset = 0;
  int updateParsFlag;
  float energy, signalEnergy, sumMagn;
  float win[ANAL], magn[BINS], noise[BINS], lmagn[BINS];
  float snrLocPost[BINS], snrLocPrior[BINS], re[BINS], im[BINS], tmpv[BINS];
  float speechProb[BINS];

  ensure_tables();
  updateParsFlag = s->modelUpdatePars[0]; /* :1065 */
  slide_and_window(s->analyzeBuf, frame, win, s->fs); /* :1068-1070 */
  energy = energy256(win, mode, 1, ANAL);
  if (energy == 0.0) return; /* :1072-1082 */
  s->blockInd++;
  forward_spectrum(win, re, im, magn, s->fs); /* :1086 */

  /* lmagn is needed three times with the same value: NoiseEstimation :228,
   * the startup fit :1096 and the flatness numerator :540. */
  for (i = 0; i < BINS; i++) lmagn[i] = (float)log(magn[i]);

  for (i = 0; i < BINS; i++) tmpv[i] = re[i] * re[i] + im[i] * im[i]; /* :1089 */
  signalEnergy = sum_bins(tmpv, mode, BINS);
  sumMagn = sum_bins(magn, mode, BINS); /* :1090 */
  signalEnergy = signalEnergy / ((float)BINS); /* :1102-1104 */
  s->signalEnergy = signalEnergy;
  s->sumMagn = sumMagn;

  /* ---- NoiseEstimation, ns_core.c:217-285 */
  if (s->updates < END_STARTUP_LONG) s->updates++;
  for (k = 0; k < SIMULT; k++) {
    offset = k * BINS;
    for (i = 0; i < BINS; i++) {
      float delta;
      if (s->density[offset + i] > 1.0)
        delta = FACTOR * 1.f / s->density[offset + i];
      else
        delta = FACTOR;
      if (lmagn[i] > s->lquantile[offset + i])
        s->lquantile[offset + i] += QUANTILE * delta / (float)(s->counter[k] + 1);
      else
        s->lquantile[offset + i] -= (1.f - QUANTILE) * delta / (float)(s->counter[k] + 1);
      if (fabs(lmagn[i] - s->lquantile[offset + i]) < WIDTH)
        s->density[offset + i] =
            ((float)s->counter[k] * s->density[offset + i] + 1.f / (2.f * WIDTH)) /
            (float)(s->counter[k] + 1);
    }
    if (s->counter[k] >= END_STARTUP_LONG) {
      s->counter[k] = 0;
      if (s->updates >= END_STARTUP_LONG)
        for (i = 0; i < BINS; i++) s->quantile[i] = (float)exp(s->lquantile[offset + i]);
    }
    s->counter[k]++;
  }
  if (s->updates < END_STARTUP_LONG) /* :275-280, offset is the last tracker's */
    for (i = 0; i < BINS; i++) s->quantile[i] = (float)exp(s->lquantile[offset + i]);
  for (i = 0; i < BINS; i++) noise[i] = s->quantile[i];

  /* ---- startup noise model, ns_core.c:1091-1100, 1109-1162 */
  if (s->blockInd < END_STARTUP_SHORT) {
    float sum_log_i = 0.f, sum_log_i_square = 0.f;
    float sum_log_magn, sum_log_i_log_magn;
    float tmpFloat1, tmpFloat2, tmpFloat3;
    float parametric_exp = 0.f, parametric_num = 0.f;
    /* data-independent sums: always in the reference's order */
    for (i = K_START_BAND; i < BINS; i++) {
      sum_log_i += g_logi[i];
      sum_log_i_square += g_logi[i] * g_logi[i];
    }
    for (i = 0; i < BINS; i++) tmpv[i] = i >= K_START_BAND ? lmagn[i] : 0.f;
    sum_log_magn = sum_bins(tmpv, mode, BINS);
    for (i = 0; i < BINS; i++) tmpv[i] = i >= K_START_BAND ? g_logi[i] * lmagn[i] : 0.f;
    sum_log_i_log_magn = sum_bins(tmpv, mode, BINS);

    s->whiteNoiseLevel += sumMagn / ((float)BINS) * s->overdrive; /* :1111 */
    tmpFloat1 = sum_log_i_square * ((float)(BINS - K_START_BAND));
    tmpFloat1 -= (sum_log_i * sum_log_i);
    tmpFloat2 = (sum_log_i_square * sum_log_magn - sum_log_i * sum_log_i_log_magn);
    tmpFloat3 = tmpFloat2 / tmpFloat1;
    if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
    s->pinkNoiseNumerator += tmpFloat3;
    tmpFloat2 = (sum_log_i * sum_log_magn);
    tmpFloat2 -= ((float)(BINS - K_START_BAND)) * sum_log_i_log_magn;
    tmpFloat3 = tmpFloat2 / tmpFloat1;
    if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
    if (tmpFloat3 > 1.f) tmpFloat3 = 1.f;
    s->pinkNoiseExp += tmpFloat3;
    if (s->pinkNoiseExp > 0.f) { /* :1136-1142 */
      parametric_num = (float)exp(s->pinkNoiseNumerator / (float)(s->blockInd + 1));
      parametric_num *= (float)(s->blockInd + 1);
      parametric_exp = s->pinkNoiseExp / (float)(s->blockInd + 1);
    }
    for (i = 0; i < BINS; i++) { /* :1143-1161 */
      if (s->pinkNoiseExp == 0.f) {
        s->parametricNoise[i] = s->whiteNoiseLevel;
      } else {
        float use_band = (float)(i < K_START_BAND ? K_START_BAND : i);
        s->parametricNoise[i] = (float)(parametric_num / pow(use_band, parametric_exp));
      }
      noise[i] *= (s->blockInd);
      tmpFloat2 = s->parametricNoise[i] * (END_STARTUP_SHORT - s->blockInd);
      noise[i] += (tmpFloat2 / (float)(s->blockInd + 1));
      noise[i] /= END_STARTUP_SHORT;
    }
  }
  if (s->blockInd < END_STARTUP_LONG) { /* :1165-1169 */
    s->featureData[5] *= s->blockInd;
    s->featureData[5] += signalEnergy;
    s->featureData[5] /= (s->blockInd + 1);
  }

  /* ---- ComputeSnr, ns_core.c:566-588 */
  for (i = 0; i < BINS; i++) {
    float previousEstimateStsa =
        s->magnPrevAnalyze[i] / (s->noisePrev[i] + 0.0001f) * s->smooth[i];
    snrLocPost[i] = 0.f;
    if (magn[i] > noise[i]) snrLocPost[i] = magn[i] / (noise[i] + 0.0001f) - 1.f;
    snrLocPrior[i] = DD_PR_SNR * previousEstimateStsa + (1.f - DD_PR_SNR) * snrLocPost[i];
  }

  /* ---- FeatureUpdate, ns_core.c:755-791 */
  { /* ComputeSpectralFlatness :523-556 (magn >= 1 so the log(0) exit is dead) */
    float num, den, spectralTmp;
    for (i = 0; i < BINS; i++) tmpv[i] = i >= 1 ? lmagn[i] : 0.f;
    num = sum_bins(tmpv, mode, BINS);
    den = s->sumMagn - magn[0];
    den = den / BINS;
    num = num / BINS;
    spectralTmp = (float)exp(num) / den;
    s->featureData[0] += SPECT_FL_TAVG * (spectralTmp - s->featureData[0]);
  }
  { /* ComputeSpectralDifference :595-634 */
    float avgPause, avgMagn, covMagnPause, varPause, varMagn, avgDiffNormMagn;
    avgPause = sum_bins(s->magnAvgPause, mode, BINS);
    avgMagn = s->sumMagn;
    avgPause = avgPause / ((float)BINS);
    avgMagn = avgMagn / ((float)BINS);
    for (i = 0; i < BINS; i++) tmpv[i] = (magn[i] - avgMagn) * (s->magnAvgPause[i] - avgPause);
    covMagnPause = sum_bins(tmpv, mode, BINS);
    for (i = 0; i < BINS; i++)
      tmpv[i] = (s->magnAvgPause[i] - avgPause) * (s->magnAvgPause[i] - avgPause);
    varPause = sum_bins(tmpv, mode, BINS);
    for (i = 0; i < BINS; i++) tmpv[i] = (magn[i] - avgMagn) * (magn[i] - avgMagn);
    varMagn = sum_bins(tmpv, mode, BINS);
    covMagnPause = covMagnPause / ((float)BINS);
    varPause = varPause / ((float)BINS);
    varMagn = varMagn / ((float)BINS);
    s->featureData[6] += s->signalEnergy;
    avgDiffNormMagn = varMagn - (covMagnPause * covMagnPause) / (varPause + 0.0001f);
    avgDiffNormMagn = (float)(avgDiffNormMagn / (s->featureData[5] + 0.0001f));
    s->featureData[4] += SPECT_DIFF_TAVG * (avgDiffNormMagn - s->featureData[4]);
  }
  if (updateParsFlag >= 1) { /* :766-790 */
    s->modelUpdatePars[3]--;
    if (s->modelUpdatePars[3] > 0) update_histograms(s);
    if (s->modelUpdatePars[3] == 0) {
      extract_parameters(s);
      s->modelUpdatePars[3] = s->modelUpdatePars[1];
      if (updateParsFlag == 1) {
        s->modelUpdatePars[0] = 0;
      } else {
        s->featureData[6] = s->featureData[6] / ((float)s->modelUpdatePars[1]);
        s->featureData[5] = 0.5f * (s->featureData[6] + s->featureData[5]);
        s->featureData[6] = 0.f;
      }
    }
  }

  /* ---- SpeechNoiseProb, ns_core.c:642-749 */
  {
    int sgnMap;
    float invLrt, gainPrior, indPrior, logLrtTimeAvgKsum;
    float indicator0, indicator1, indicator2, tmpFloat1, widthPrior;
    const float widthPrior0 = WIDTH_PR_MAP, widthPrior1 = 2.f * WIDTH_PR_MAP,
                widthPrior2 = 2.f * WIDTH_PR_MAP;
    const float threshPrior0 = s->priorModelPars[0], threshPrior1 = s->priorModelPars[1],
                threshPrior2 = s->priorModelPars[3];
    sgnMap = (int)(s->priorModelPars[2]);
    for (i = 0; i < BINS; i++) { /* :676-683 */
      float t1 = 1.f + 2.f * snrLocPrior[i];
      float t2 = 2.f * snrLocPrior[i] / (t1 + 0.0001f);
      float besselTmp = (snrLocPost[i] + 1.f) * t2;
      s->logLrtTimeAvg[i] += LRT_TAVG * (besselTmp - (float)log(t1) - s->logLrtTimeAvg[i]);
    }
    logLrtTimeAvgKsum = sum_bins(s->logLrtTimeAvg, mode, BINS);
    logLrtTimeAvgKsum = (float)logLrtTimeAvgKsum / (BINS);
    s->featureData[3] = logLrtTimeAvgKsum;
    widthPrior = widthPrior0; /* :690-698 */
    if (logLrtTimeAvgKsum < threshPrior0) widthPrior = widthPrior1;
    indicator0 = 0.5f * ((float)tanh(widthPrior * (logLrtTimeAvgKsum - threshPrior0)) + 1.f);
    tmpFloat1 = s->featureData[0]; /* :701-714 */
    widthPrior = widthPrior0;
    if (sgnMap == 1 && (tmpFloat1 > threshPrior1)) widthPrior = widthPrior1;
    if (sgnMap == -1 && (tmpFloat1 < threshPrior1)) widthPrior = widthPrior1;
    indicator1 =
        0.5f * ((float)tanh((float)sgnMap * widthPrior * (threshPrior1 - tmpFloat1)) + 1.f);
    tmpFloat1 = s->featureData[4]; /* :717-725 */
    widthPrior = widthPrior0;
    if (tmpFloat1 < threshPrior2) widthPrior = widthPrior2;
    indicator2 = 0.5f * ((float)tanh(widthPrior * (tmpFloat1 - threshPrior2)) + 1.f);
    indPrior = s->priorModelPars[4] * indicator0 + s->priorModelPars[5] * indicator1 +
               s->priorModelPars[6] * indicator2; /* :728-729 */
    s->priorSpeechProb += PRIOR_UPDATE * (indPrior - s->priorSpeechProb);
    if (s->priorSpeechProb > 1.f) s->priorSpeechProb = 1.f;
    if (s->priorSpeechProb < 0.01f) s->priorSpeechProb = 0.01f;
    gainPrior = (1.f - s->priorSpeechProb) / (s->priorSpeechProb + 0.0001f); /* :743 */
    for (i = 0; i < BINS; i++) {
      invLrt = (float)exp(-s->logLrtTimeAvg[i]);
      invLrt = (float)gainPrior * invLrt;
      speechProb[i] = 1.f / (1.f + invLrt);
    }
  }

  /* ---- UpdateNoiseEstimate, ns_core.c:800-846 */
  {
    float gammaNoiseTmp = NOISE_UPDATE;
    for (i = 0; i < BINS; i++) {
      float probSpeech = speechProb[i];
      float probNonSpeech = 1.f - probSpeech;
      float gammaNoiseOld;
      float noiseUpdateTmp =
          gammaNoiseTmp * s->noisePrev[i] +
          (1.f - gammaNoiseTmp) * (probNonSpeech * magn[i] + probSpeech * s->noisePrev[i]);
      gammaNoiseOld = gammaNoiseTmp;
      gammaNoiseTmp = NOISE_UPDATE;
      if (probSpeech > PROB_RANGE) gammaNoiseTmp = SPEECH_UPDATE;
      if (probSpeech < PROB_RANGE)
        s->magnAvgPause[i] += GAMMA_PAUSE * (magn[i] - s->magnAvgPause[i]);
      if (gammaNoiseTmp == gammaNoiseOld) {
        noise[i] = noiseUpdateTmp;
      } else {
        noise[i] = gammaNoiseTmp * s->noisePrev[i] +
                   (1.f - gammaNoiseTmp) *
                       (probNonSpeech * magn[i] + probSpeech * s->noisePrev[i]);
        if (noiseUpdateTmp < noise[i]) noise[i] = noiseUpdateTmp;
      }
    }
  }
  memcpy(s->speechProb, speechProb, sizeof speechProb);
  memcpy(s->noise, noise, sizeof noise);             /* :1179 */
  memcpy(s->magnPrevAnalyze, magn, sizeof magn);     /* :1180 */
}

/* --------------------------------------------------------------- synthesis */

static float sat16(float x) { /* WEBRTC_SPL_SAT(32767, x, -32768), ns_core.c:1357-1359 */
  return x > 32767 ? 32767 : (x < -32768 ? -32768 : x);
}

/* returns 0 for the zero-energy early exit (ns_core.c:1239-1264), 1 otherwise */
static int process_low(AspNsState* s, const float* in, float* out, int mode) {
  GEO(s);
  int i;
  float energy1, energy2, gain, factor, factor1, factor2;
  float fout[BLOCKL], win[ANAL], magn[BINS], theFilter[BINS], re[BINS], im[BINS];

  ensure_tables();
  slide_and_window(s->dataBuf, in, win, s->fs); /* :1225, :1237 */
  energy1 = energy256(win, mode, 1, ANAL);
  if (energy1 == 0.0) { /* :1239-1264 */
    for (i = 0; i < BLOCKL; i++) fout[i] = s->syntBuf[i];
    slide_and_window(s->syntBuf, NULL, NULL, s->fs);
    for (i = 0; i < BLOCKL; ++i) out[i] = sat16(fout[i]);
    return 0;
  }
  forward_spectrum(win, re, im, magn, s->fs); /* :1266 */
  if (s->blockInd < END_STARTUP_SHORT) /* :1268-1272 */
    for (i = 0; i < BINS; i++) s->initMagnEst[i] += magn[i];

  for (i = 0; i < BINS; i++) { /* ComputeDdBasedWienerFilter :985-1007 */
    float previousEstimateStsa =
        s->magnPrevProcess[i] / (s->noisePrev[i] + 0.0001f) * s->smooth[i];
    float currentEstimateStsa = 0.f, snrPrior;
    if (magn[i] > s->noise[i]) currentEstimateStsa = magn[i] / (s->noise[i] + 0.0001f) - 1.f;
    snrPrior = DD_PR_SNR * previousEstimateStsa + (1.f - DD_PR_SNR) * currentEstimateStsa;
    theFilter[i] = snrPrior / (s->overdrive + snrPrior);
  }
  for (i = 0; i < BINS; i++) { /* :1276-1307 */
    if (theFilter[i] < s->denoiseBound) theFilter[i] = s->denoiseBound;
    if (theFilter[i] > 1.f) theFilter[i] = 1.f;
    if (s->blockInd < END_STARTUP_SHORT) {
      float tmp = (s->initMagnEst[i] - s->overdrive * s->parametricNoise[i]);
      tmp /= (s->initMagnEst[i] + 0.0001f);
      if (tmp < s->denoiseBound) tmp = s->denoiseBound;
      if (tmp > 1.f) tmp = 1.f;
      theFilter[i] *= (s->blockInd);
      tmp *= (END_STARTUP_SHORT - s->blockInd);
      theFilter[i] += tmp;
      theFilter[i] /= (END_STARTUP_SHORT);
    }
    s->smooth[i] = theFilter[i];
    re[i] *= s->smooth[i];
    im[i] *= s->smooth[i];
  }
  memcpy(s->magnPrevProcess, magn, sizeof magn);       /* :1309 */
  memcpy(s->noisePrev, s->noise, sizeof s->noise);     /* :1310 */

  win[0] = re[0]; /* IFFT(), ns_core.c:923-944 */
  win[1] = re[BINS - 1];
  for (i = 1; i < BINS - 1; ++i) {
    win[2 * i] = re[i];
    win[2 * i + 1] = im[i];
  }
  rdft_n(win, ANAL, -1);
  for (i = 0; i < ANAL; ++i) win[i] *= 2.f / ANAL;

  factor = 1.f; /* :1315-1342 */
  if (s->gainmap == 1 && s->blockInd > END_STARTUP_LONG) {
    factor1 = 1.f;
    factor2 = 1.f;
    energy2 = energy256(win, mode, 0, ANAL);
    gain = (float)sqrt(energy2 / (energy1 + 1.f));
    if (gain > B_LIM) {
      factor1 = 1.f + 1.3f * (gain - B_LIM);
      if (gain * factor1 > 1.f) factor1 = 1.f / gain;
    }
    if (gain < B_LIM) {
      if (gain <= s->denoiseBound) gain = s->denoiseBound;
      factor2 = 1.f - 0.3f * (B_LIM - gain);
    }
    factor = s->priorSpeechProb * factor1 + (1.f - s->priorSpeechProb) * factor2;
  }
  for (i = 0; i < ANAL; ++i) win[i] = (s->fs == 8000 ? g_window8 : g_window)[i] * win[i]; /* :1344 */
  for (i = 0; i < ANAL; i++) s->syntBuf[i] += factor * win[i]; /* :1347-1349 */
  for (i = 0; i < BLOCKL; i++) fout[i] = s->syntBuf[i];
  slide_and_window(s->syntBuf, NULL, NULL, s->fs); /* :1355 */
  for (i = 0; i < BLOCKL; ++i) out[i] = sat16(fout[i]);
  return 1;
}

void asp_ns_oracle_process(AspNsState* s, const float* in, float* out, int mode) {
  (void)process_low(s, in, out, mode);
}

void asp_ns_oracle_process_bands(AspNsState* s, AspNsHbState* hb, const float* in_low,
                                 const float* in_high, int num_high, float* out_low,
                                 float* out_high, int mode) {
  GEO(s);
  int i, j, live;
  const int deltaBweHB = BINS / 4, deltaGainHB = BINS / 4; /* :1221-1223 */
  float avgProbSpeechHB, avgProbSpeechHBTmp, avgFilterGainHB, gainModHB, gainTimeDomainHB;
  float sumMagnAnalyze, sumMagnProcess;
  const float decayBweHB = 1.0, gainMapParHB = 1.0; /* :1202-1203 */
  for (i = 0; i < num_high; ++i) /* UpdateBuffer of the high bands, :1227-1235 */
    slide_and_window(hb->dataBufHB[i], in_high + (size_t)i * BLOCKL, NULL, s->fs);
  live = process_low(s, in_low, out_low, mode);
  if (!live) { /* :1252-1261 */
    for (i = 0; i < num_high; ++i)
      for (j = 0; j < BLOCKL; ++j) out_high[(size_t)i * BLOCKL + j] = sat16(hb->dataBufHB[i][j]);
    return;
  }
  /* :1362-1414 */
  avgProbSpeechHB = 0.0;
  for (i = BINS - deltaBweHB - 1; i < BINS - 1; i++) avgProbSpeechHB += s->speechProb[i];
  avgProbSpeechHB = avgProbSpeechHB / ((float)deltaBweHB);
  sumMagnAnalyze = 0;
  sumMagnProcess = 0;
  for (i = 0; i < BINS; ++i) {
    sumMagnAnalyze += s->magnPrevAnalyze[i];
    sumMagnProcess += s->magnPrevProcess[i];
  }
  avgProbSpeechHB *= sumMagnProcess / sumMagnAnalyze;
  avgFilterGainHB = 0.0;
  for (i = BINS - deltaGainHB - 1; i < BINS - 1; i++) avgFilterGainHB += s->smooth[i];
  avgFilterGainHB = avgFilterGainHB / ((float)(deltaGainHB));
  avgProbSpeechHBTmp = 2.f * avgProbSpeechHB - 1.f;
  gainModHB = 0.5f * (1.f + (float)tanh(gainMapParHB * avgProbSpeechHBTmp));
  gainTimeDomainHB = 0.5f * gainModHB + 0.5f * avgFilterGainHB;
  if (avgProbSpeechHB >= 0.5f) gainTimeDomainHB = 0.25f * gainModHB + 0.75f * avgFilterGainHB;
  gainTimeDomainHB = gainTimeDomainHB * decayBweHB;
  if (gainTimeDomainHB < s->denoiseBound) gainTimeDomainHB = s->denoiseBound;
  if (gainTimeDomainHB > 1.f) gainTimeDomainHB = 1.f;
  for (i = 0; i < num_high; ++i)
    for (j = 0; j < BLOCKL; j++)
      out_high[(size_t)i * BLOCKL + j] = sat16(gainTimeDomainHB * hb->dataBufHB[i][j]);
}

/* ------------------------------------------------------------ batch helper */

void asp_ns_oracle_run(AspNsState* states, int num_streams, const float* in,
                       float* out, int num_frames, int mode) {
  GEO(&states[0]); /* one geometry per batch */
  for (int f = 0; f < num_frames; ++f)
    for (int st = 0; st < num_streams; ++st) {
      const float* x = in + ((size_t)f * num_streams + st) * BLOCKL;
      float* y = out + ((size_t)f * num_streams + st) * BLOCKL;
      float tmp[BLOCKL];
      memcpy(tmp, x, sizeof tmp); /* in == out aliasing is legal */
      asp_ns_oracle_analyze(&states[st], tmp, mode);
      asp_ns_oracle_process(&states[st], tmp, y, mode);
    }
}

typedef struct {
  AspNsState* states;
  int s0, s1, num_streams, num_frames, mode;
  const float* in;
  float* out;
} Shard;

static void* shard_main(void* p) {
  Shard* sh = (Shard*)p;
  GEO(&sh->states[0]);
  for (int f = 0; f < sh->num_frames; ++f)
    for (int st = sh->s0; st < sh->s1; ++st) {
      const float* x = sh->in + ((size_t)f * sh->num_streams + st) * BLOCKL;
      float* y = sh->out + ((size_t)f * sh->num_streams + st) * BLOCKL;
      float tmp[BLOCKL];
      memcpy(tmp, x, sizeof tmp);
      asp_ns_oracle_analyze(&sh->states[st], tmp, sh->mode);
      asp_ns_oracle_process(&sh->states[st], tmp, y, sh->mode);
    }
  return NULL;
}

void asp_ns_oracle_run_mt(AspNsState* states, int num_streams, const float* in,
                          float* out, int num_frames, int mode, int threads) {
  if (threads < 1) threads = 1;
  if (threads > num_streams) threads = num_streams;
  ensure_tables();
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  Shard* sh = (Shard*)malloc(sizeof(Shard) * (size_t)threads);
  for (int t = 0; t < threads; ++t) {
    sh[t].states = states;
    sh[t].s0 = (int)((long long)num_streams * t / threads);
    sh[t].s1 = (int)((long long)num_streams * (t + 1) / threads);
    sh[t].num_streams = num_streams;
    sh[t].num_frames = num_frames;
    sh[t].mode = mode;
    sh[t].in = in;
    sh[t].out = out;
    pthread_create(&th[t], NULL, shard_main, &sh[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th);
  free(sh);
}

/* The host libm idioms of the reference, (float)fn((double)x), applied in place:
 * fn 1 = log, 2 = exp, 3 = tanh.  Used to check the device math against glibc. */
void asp_oracle_libm_f32(int fn, float* data, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const double x = (double)data[i];
    data[i] = (float)(fn == 1 ? log(x) : (fn == 2 ? exp(x) : tanh(x)));
  }
}
