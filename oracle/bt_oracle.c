/*
 * bt_oracle.c -- CPU restatement of the reference's block-thresholding denoiser
 * (Denoise/BlockThresholding/src/audioDenoiseBlockTreshold.c) and of the part of
 * kiss_fft it uses (common/kiss_fft/kiss_fft.c, kiss_fftr.c: every radix --
 * 4, 2, 3, 5 and the generic butterfly -- so any even window length).
 * TEST INFRASTRUCTURE ONLY -- see bt_oracle.h.
 *
 * PARITY UNPINNED for the FFT internals: common/kiss_fft/_kiss_fft_guts.h is
 * missing from the reference snapshot, so kiss_fft -- and with it the whole
 * BlockThresholding program -- cannot be compiled here without writing a
 * stand-in header, and the reference ships no expected outputs for it
 * (unittest_fft.cpp / unittest_real_fft.cpp only print).  The complex
 * multiply / add macros are restated from their uses in kiss_fft.c:21-90 and
 * kiss_fftr.c:92-157 (upstream kissfft 1.3.0 semantics: C_MUL is the plain
 * four-multiply product, HALF_OF(x) = x*.5, C_FIXDIV a no-op in float); the
 * restatement is anchored on numpy rfft agreement and round-trip identities
 * (tests/test_bt_oracle.py).  Everything above the FFT follows
 * audioDenoiseBlockTreshold.c line by line in float arithmetic.
 */
#include "bt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NT ASP_BT_NBLK_TIME /* 8 */
#define NF ASP_BT_NBLK_FREQ /* 16 */
#define POW2(x) ((x) * (x))

typedef struct {
  float r, i;
} cpx;

/* ---------------------------------------------------------------- kiss_fft */

typedef struct {
  int nfft, inverse;
  int factors[32];
  cpx* tw;
} KissCfg;

static void kiss_cfg_init(KissCfg* st, int nfft, int inverse) { /* kiss_fft.c:339-368 */
  st->nfft = nfft;
  st->inverse = inverse;
  st->tw = (cpx*)malloc(sizeof(cpx) * (size_t)nfft);
  for (int i = 0; i < nfft; ++i) {
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    double phase = -2 * pi * i / nfft;
    if (inverse) phase *= -1;
    st->tw[i].r = (float)cos(phase);
    st->tw[i].i = (float)sin(phase);
  }
  /* kf_factor, kiss_fft.c:308-330: fours first, then twos, then the odd primes in turn; once the
   * trial factor passes floor(sqrt(n)) what is left is prime and taken whole */
  int n = nfft, p = 4, k = 0;
  const double floor_sqrt = floor(sqrt((double)nfft));
  do {
    while (n % p) {
      if (p == 4)
        p = 2;
      else if (p == 2)
        p = 3;
      else
        p += 2;
      if (p > floor_sqrt) p = n;
    }
    n /= p;
    st->factors[k++] = p;
    st->factors[k++] = n;
  } while (n > 1);
}

static void bfly2(cpx* F, int fstride, const KissCfg* st, int m) { /* kiss_fft.c:21-42 */
  cpx* F2 = F + m;
  const cpx* tw = st->tw;
  for (int j = 0; j < m; ++j) {
    cpx t;
    t.r = F2[j].r * tw->r - F2[j].i * tw->i;
    t.i = F2[j].r * tw->i + F2[j].i * tw->r;
    tw += fstride;
    F2[j].r = F[j].r - t.r;
    F2[j].i = F[j].i - t.i;
    F[j].r += t.r;
    F[j].i += t.i;
  }
}

static void bfly4(cpx* F, int fstride, const KissCfg* st, int m) { /* kiss_fft.c:44-90 */
  const cpx *tw1 = st->tw, *tw2 = st->tw, *tw3 = st->tw;
  const int m2 = 2 * m, m3 = 3 * m;
  for (int k = 0; k < m; ++k, ++F) {
    cpx s0, s1, s2, s3, s4, s5;
    s0.r = F[m].r * tw1->r - F[m].i * tw1->i;
    s0.i = F[m].r * tw1->i + F[m].i * tw1->r;
    s1.r = F[m2].r * tw2->r - F[m2].i * tw2->i;
    s1.i = F[m2].r * tw2->i + F[m2].i * tw2->r;
    s2.r = F[m3].r * tw3->r - F[m3].i * tw3->i;
    s2.i = F[m3].r * tw3->i + F[m3].i * tw3->r;
    s5.r = F->r - s1.r;
    s5.i = F->i - s1.i;
    F->r += s1.r;
    F->i += s1.i;
    s3.r = s0.r + s2.r;
    s3.i = s0.i + s2.i;
    s4.r = s0.r - s2.r;
    s4.i = s0.i - s2.i;
    F[m2].r = F->r - s3.r;
    F[m2].i = F->i - s3.i;
    tw1 += fstride;
    tw2 += fstride * 2;
    tw3 += fstride * 3;
    F->r += s3.r;
    F->i += s3.i;
    if (st->inverse) {
      F[m].r = s5.r - s4.i;
      F[m].i = s5.i + s4.r;
      F[m3].r = s5.r + s4.i;
      F[m3].i = s5.i - s4.r;
    } else {
      F[m].r = s5.r + s4.i;
      F[m].i = s5.i - s4.r;
      F[m3].r = s5.r - s4.i;
      F[m3].i = s5.i + s4.r;
    }
  }
}

static cpx cmul(cpx a, cpx b) { /* C_MUL: the plain four-multiply product */
  cpx m;
  m.r = a.r * b.r - a.i * b.i;
  m.i = a.r * b.i + a.i * b.r;
  return m;
}

static void bfly3(cpx* F, int fstride, const KissCfg* st, int m) { /* kiss_fft.c:92-136 */
  const int m2 = 2 * m;
  const cpx *tw1 = st->tw, *tw2 = st->tw;
  const cpx epi3 = st->tw[fstride * m];
  for (int k = 0; k < m; ++k, ++F) {
    const cpx s1 = cmul(F[m], *tw1), s2 = cmul(F[m2], *tw2);
    cpx s3, s0;
    s3.r = s1.r + s2.r;
    s3.i = s1.i + s2.i;
    s0.r = s1.r - s2.r;
    s0.i = s1.i - s2.i;
    tw1 += fstride;
    tw2 += fstride * 2;
    F[m].r = F->r - s3.r * .5f; /* HALF_OF */
    F[m].i = F->i - s3.i * .5f;
    s0.r *= epi3.i; /* C_MULBYSCALAR */
    s0.i *= epi3.i;
    F->r += s3.r;
    F->i += s3.i;
    F[m2].r = F[m].r + s0.i;
    F[m2].i = F[m].i - s0.r;
    F[m].r -= s0.i;
    F[m].i += s0.r;
  }
}

static void bfly5(cpx* F, int fstride, const KissCfg* st, int m) { /* kiss_fft.c:138-197 */
  const cpx* tw = st->tw;
  const cpx ya = tw[fstride * m], yb = tw[fstride * 2 * m];
  cpx *F0 = F, *F1 = F + m, *F2 = F + 2 * m, *F3 = F + 3 * m, *F4 = F + 4 * m;
  for (int u = 0; u < m; ++u) {
    cpx s[13];
    s[0] = *F0;
    s[1] = cmul(*F1, tw[u * fstride]);
    s[2] = cmul(*F2, tw[2 * u * fstride]);
    s[3] = cmul(*F3, tw[3 * u * fstride]);
    s[4] = cmul(*F4, tw[4 * u * fstride]);
    s[7].r = s[1].r + s[4].r;
    s[7].i = s[1].i + s[4].i;
    s[10].r = s[1].r - s[4].r;
    s[10].i = s[1].i - s[4].i;
    s[8].r = s[2].r + s[3].r;
    s[8].i = s[2].i + s[3].i;
    s[9].r = s[2].r - s[3].r;
    s[9].i = s[2].i - s[3].i;
    F0->r += s[7].r + s[8].r;
    F0->i += s[7].i + s[8].i;
    s[5].r = s[0].r + s[7].r * ya.r + s[8].r * yb.r;
    s[5].i = s[0].i + s[7].i * ya.r + s[8].i * yb.r;
    s[6].r = s[10].i * ya.i + s[9].i * yb.i;
    s[6].i = -(s[10].r * ya.i) - s[9].r * yb.i;
    F1->r = s[5].r - s[6].r;
    F1->i = s[5].i - s[6].i;
    F4->r = s[5].r + s[6].r;
    F4->i = s[5].i + s[6].i;
    s[11].r = s[0].r + s[7].r * yb.r + s[8].r * ya.r;
    s[11].i = s[0].i + s[7].i * yb.r + s[8].i * ya.r;
    s[12].r = -(s[10].i * yb.i) + s[9].i * ya.i;
    s[12].i = s[10].r * yb.i - s[9].r * ya.i;
    F2->r = s[11].r + s[12].r;
    F2->i = s[11].i + s[12].i;
    F3->r = s[11].r - s[12].r;
    F3->i = s[11].i - s[12].i;
    ++F0, ++F1, ++F2, ++F3, ++F4;
  }
}

static void bfly_generic(cpx* F, int fstride, const KissCfg* st, int m, int p) { /* kiss_fft.c:199-235 */
  cpx* scratch = (cpx*)malloc(sizeof(cpx) * (size_t)p);
  for (int u = 0; u < m; ++u) {
    int k = u;
    for (int q1 = 0; q1 < p; ++q1, k += m) scratch[q1] = F[k];
    k = u;
    for (int q1 = 0; q1 < p; ++q1, k += m) {
      int twidx = 0;
      F[k] = scratch[0];
      for (int q = 1; q < p; ++q) {
        twidx += fstride * k;
        if (twidx >= st->nfft) twidx -= st->nfft;
        const cpx t = cmul(scratch[q], st->tw[twidx]);
        F[k].r += t.r;
        F[k].i += t.i;
      }
    }
  }
  free(scratch);
}

static void kf_work(cpx* Fout, const cpx* f, int fstride, const int* factors,
                    const KissCfg* st) { /* kiss_fft.c:237-302 */
  const int p = factors[0], m = factors[1];
  if (m == 1) {
    for (int k = 0; k < p; ++k) Fout[k] = f[(size_t)k * fstride];
  } else {
    for (int k = 0; k < p; ++k) kf_work(Fout + (size_t)k * m, f + (size_t)k * fstride, fstride * p, factors + 2, st);
  }
  switch (p) {
    case 2: bfly2(Fout, fstride, st, m); break;
    case 3: bfly3(Fout, fstride, st, m); break;
    case 4: bfly4(Fout, fstride, st, m); break;
    case 5: bfly5(Fout, fstride, st, m); break;
    default: bfly_generic(Fout, fstride, st, m, p); break;
  }
}

typedef struct {
  int n; /* real length */
  KissCfg fwd, inv;
  cpx *sup_fwd, *sup_inv, *tmp;
} KissR;

static void kissr_init(KissR* k, int n) { /* kiss_fftr.c:27-65 */
  const int nc = n / 2;
  k->n = n;
  kiss_cfg_init(&k->fwd, nc, 0);
  kiss_cfg_init(&k->inv, nc, 1);
  k->sup_fwd = (cpx*)malloc(sizeof(cpx) * (size_t)(nc / 2));
  k->sup_inv = (cpx*)malloc(sizeof(cpx) * (size_t)(nc / 2));
  k->tmp = (cpx*)malloc(sizeof(cpx) * (size_t)nc);
  for (int i = 0; i < nc / 2; ++i) {
    double phase = -3.14159265358979323846264338327 * ((double)(i + 1) / nc + .5);
    k->sup_fwd[i].r = (float)cos(phase);
    k->sup_fwd[i].i = (float)sin(phase);
    phase *= -1;
    k->sup_inv[i].r = (float)cos(phase);
    k->sup_inv[i].i = (float)sin(phase);
  }
}

static void kissr_free(KissR* k) {
  free(k->fwd.tw);
  free(k->inv.tw);
  free(k->sup_fwd);
  free(k->sup_inv);
  free(k->tmp);
}

static void kiss_fftr_fwd(KissR* st, const float* timedata, cpx* freq) { /* kiss_fftr.c:67-121 */
  const int nc = st->n / 2;
  kf_work(st->tmp, (const cpx*)timedata, 1, st->fwd.factors, &st->fwd);
  const float tdr = st->tmp[0].r, tdi = st->tmp[0].i;
  freq[0].r = tdr + tdi;
  freq[nc].r = tdr - tdi;
  freq[nc].i = freq[0].i = 0;
  for (int k = 1; k <= nc / 2; ++k) {
    cpx fpk = st->tmp[k], fpnk, f1k, f2k, tw;
    fpnk.r = st->tmp[nc - k].r;
    fpnk.i = -st->tmp[nc - k].i;
    f1k.r = fpk.r + fpnk.r;
    f1k.i = fpk.i + fpnk.i;
    f2k.r = fpk.r - fpnk.r;
    f2k.i = fpk.i - fpnk.i;
    tw.r = f2k.r * st->sup_fwd[k - 1].r - f2k.i * st->sup_fwd[k - 1].i;
    tw.i = f2k.r * st->sup_fwd[k - 1].i + f2k.i * st->sup_fwd[k - 1].r;
    freq[k].r = (float)((f1k.r + tw.r) * .5);
    freq[k].i = (float)((f1k.i + tw.i) * .5);
    freq[nc - k].r = (float)((f1k.r - tw.r) * .5);
    freq[nc - k].i = (float)((tw.i - f1k.i) * .5);
  }
}

static void kiss_fftr_inv(KissR* st, const cpx* freq, float* timedata) { /* kiss_fftr.c:123-159 */
  const int nc = st->n / 2;
  st->tmp[0].r = freq[0].r + freq[nc].r;
  st->tmp[0].i = freq[0].r - freq[nc].r;
  for (int k = 1; k <= nc / 2; ++k) {
    cpx fk = freq[k], fnkc, fek, fok, tmp;
    fnkc.r = freq[nc - k].r;
    fnkc.i = -freq[nc - k].i;
    fek.r = fk.r + fnkc.r;
    fek.i = fk.i + fnkc.i;
    tmp.r = fk.r - fnkc.r;
    tmp.i = fk.i - fnkc.i;
    fok.r = tmp.r * st->sup_inv[k - 1].r - tmp.i * st->sup_inv[k - 1].i;
    fok.i = tmp.r * st->sup_inv[k - 1].i + tmp.i * st->sup_inv[k - 1].r;
    st->tmp[k].r = fek.r + fok.r;
    st->tmp[k].i = fek.i + fok.i;
    st->tmp[nc - k].r = fek.r - fok.r;
    st->tmp[nc - k].i = fek.i - fok.i;
    st->tmp[nc - k].i *= -1;
  }
  kf_work((cpx*)timedata, st->tmp, 1, st->inv.factors, &st->inv);
}

/* ------------------------------------------------------------------ handle */

struct BtOracle {
  int win, half, macro;
  int have; /* have_nblk_time */
  float* hann;
  float sigma_h; /* sigma_hanning_noise */
  float *inbuf, *inbuf_win, *outbuf;
  cpx *coef, *thre; /* [8][win/2+1] */
  KissR fft;
};

static cpx* row(cpx* base, const BtOracle* h, int t) { return base + (size_t)t * (h->win / 2 + 1); }

BtOracle* bt_oracle_create(int win_size) {
  if (win_size < 4 || win_size > 2048 || (win_size & 1)) return NULL; /* the state structs carry half <= 1024 */
  BtOracle* h = (BtOracle*)calloc(1, sizeof *h);
  h->win = win_size;
  h->half = win_size / 2;
  h->macro = h->half * NT;
  h->hann = (float*)malloc(sizeof(float) * (size_t)win_size);
  for (int i = 0; i < h->half; i++) { /* make_hanning_window, .c:70-77: symmetric Hann */
    h->hann[i] = (float)(0.5 - 0.5 * cos(2 * M_PI * i / (win_size - 1)));
    h->hann[win_size - 1 - i] = h->hann[i];
  }
  {
    float sigma_noise = (float)0.047;                 /* .c:111 */
    h->sigma_h = (float)(sigma_noise * sqrt(0.375));  /* .c:112 */
  }
  h->inbuf = (float*)calloc((size_t)win_size, sizeof(float));
  h->inbuf_win = (float*)calloc((size_t)win_size, sizeof(float));
  h->outbuf = (float*)calloc((size_t)(h->macro + h->half), sizeof(float));
  h->coef = (cpx*)calloc((size_t)NT * (win_size / 2 + 1), sizeof(cpx));
  h->thre = (cpx*)calloc((size_t)NT * (win_size / 2 + 1), sizeof(cpx));
  kissr_init(&h->fft, win_size);
  return h;
}

void bt_oracle_free(BtOracle* h) {
  if (!h) return;
  free(h->hann);
  free(h->inbuf);
  free(h->inbuf_win);
  free(h->outbuf);
  free(h->coef);
  free(h->thre);
  kissr_free(&h->fft);
  free(h);
}

void bt_oracle_reset(BtOracle* h) { /* .c:239-253 */
  h->have = 0;
  memset(h->inbuf, 0, sizeof(float) * (size_t)h->win);
  memset(h->outbuf, 0, sizeof(float) * (size_t)(h->macro + h->half));
}

int bt_oracle_win(const BtOracle* h) { return h->win; }

void bt_oracle_export(const BtOracle* h, AspBtState* s) {
  memset(s, 0, sizeof *s);
  s->win_size = h->win;
  memcpy(s->inbuf_tail, h->inbuf + h->half, sizeof(float) * (size_t)h->half);
  memcpy(s->out_tail, h->outbuf + h->macro, sizeof(float) * (size_t)h->half);
}

void bt_oracle_import(BtOracle* h, const AspBtState* s) {
  memcpy(h->inbuf + h->half, s->inbuf_tail, sizeof(float) * (size_t)h->half);
  memcpy(h->outbuf + h->macro, s->out_tail, sizeof(float) * (size_t)h->half);
  h->have = 0;
}

/* ------------------------------------------------------------------- core */

static const float m_lambda[3][5] = {{1.5, 1.8, 2, 2.5, 2.5},   /* .c:11-13 */
                                     {1.8, 2, 2.5, 3.5, 3.5},
                                     {2, 2.5, 3.5, 4.7, 4.7}};

/* power_STFT over rows 0..7 of one column (.c:303-319) */
static float column_power(const BtOracle* h, cpx* base, int col) {
  float sum = 0.0;
  for (int t = 0; t < NT; ++t) {
    float r = row(base, h, t)[col].r, i = row(base, h, t)[col].i;
    sum += POW2(r) + POW2(i);
  }
  return sum;
}

/* blockThreshold_adaptive_block (.c:354-419) + blockTreshold_compute_thre (.c:421-454)
 * for macro-column m (bins 1+16m .. 16+16m). */
static void macro_column(BtOracle* h, int m, int* seg_out) {
  cpx blk[NT][NF];
  float nrm[NT][NF]; /* real parts of stft_coef_block_norm */
  float SURE[3][5];
  const float norm = (float)(sqrt(2.0) / (sqrt(h->win) * (h->sigma_h)));
  const int base = 1 + m * NF;
  for (int t = 0; t < NT; ++t)
    for (int i = 0; i < NF; ++i) {
      blk[t][i] = row(h->coef, h, t)[base + i];
      nrm[t][i] = blk[t][i].r * norm;
    }
  for (int T = 0; T < 3; T++) {
    const int TT = NT >> T;
    for (int F = 0; F < 5; F++) {
      const int FF = NF >> F;
      const float lambda = m_lambda[T][F];
      float SURE_real = 0.0;
      const float size_blk = (float)(TT * FF);
      const float temp = POW2(lambda) * POW2(size_blk) - 2 * lambda * size_blk * (size_blk - 2);
      for (int ii = 0; ii < (1 << T); ii++)
        for (int jj = 0; jj < (1 << F); jj++) {
          float energy_real = 0.0; /* energy_real_STFT, .c:322-338: real parts only */
          for (int r = TT * ii; r <= TT * (ii + 1) - 1; r++)
            for (int c = FF * jj; c <= FF * (jj + 1) - 1; c++) energy_real += POW2(nrm[r][c]);
          SURE_real += size_blk + temp / energy_real * (energy_real > lambda * size_blk) +
                       (energy_real - 2 * size_blk) * (energy_real <= lambda * size_blk);
        }
      SURE[T][F] = SURE_real;
    }
  }
  float min_SURE = SURE[0][0];
  int seg_time = 0, seg_freq = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 5; j++)
      if (SURE[i][j] < min_SURE) {
        min_SURE = SURE[i][j];
        seg_time = i;
        seg_freq = j;
      }
  if (seg_out) {
    seg_out[2 * m] = seg_time;
    seg_out[2 * m + 1] = seg_freq;
  }
  /* compute_thre */
  const int TT = NT >> seg_time, FF = NF >> seg_freq;
  const float lambda = m_lambda[seg_time][seg_freq];
  for (int ii = 0; ii < (1 << seg_time); ii++)
    for (int jj = 0; jj < (1 << seg_freq); jj++) {
      float a = (float)(lambda * TT * FF * pow(h->sigma_h, 2) * (h->win));
      float power = 0.0;
      for (int r = TT * ii; r <= TT * (ii + 1) - 1; r++)
        for (int c = FF * jj; c <= FF * (jj + 1) - 1; c++)
          power += POW2(blk[r][c].r) + POW2(blk[r][c].i);
      a = (float)(1.0 - a / power);
      a = a * (a > 0);
      for (int kk = 0; kk < TT; kk++)
        for (int ww = 0; ww < FF; ww++) {
          const int r = ii * TT + kk, c = jj * FF + ww;
          row(h->thre, h, r)[base + c].r = blk[r][c].r * a;
          row(h->thre, h, r)[base + c].i = blk[r][c].i * a;
        }
    }
}

static void bt_core(BtOracle* h, int* seg_out) { /* blockThreshold_core, .c:488-539 */
  const float L_pi = 8.0, Lambda_pi = 2.5;
  const int ncol = (h->win - 1) / 2 / NF;
  float a;
  a = 1 - (Lambda_pi * L_pi * POW2(h->sigma_h) * (h->win)) / column_power(h, h->coef, 0);
  if (a < 0) a = 0;
  for (int t = 0; t < NT; ++t) {
    row(h->thre, h, t)[0].r = row(h->coef, h, t)[0].r * a;
    row(h->thre, h, t)[0].i = row(h->coef, h, t)[0].i * a;
  }
  for (int m = 0; m < ncol; m++) macro_column(h, m, seg_out);
  for (int i = 1 + ncol * NF; i < h->win / 2 + 1; i++) { /* .c:518-532 */
    a = Lambda_pi * L_pi * POW2(h->sigma_h) * (h->win);
    a = 1 - a / column_power(h, h->coef, i);
    if (a < 0) a = 0;
    for (int t = 0; t < NT; ++t) {
      row(h->thre, h, t)[i].r = row(h->coef, h, t)[i].r * a;
      row(h->thre, h, t)[i].i = row(h->coef, h, t)[i].i * a;
    }
  }
  for (int t = 0; t < NT; t++) /* blockThreshold_wiener, .c:469-486: Nyquist untouched */
    for (int f = 0; f < (h->win + 1) / 2; f++) {
      float r = row(h->thre, h, t)[f].r, i = row(h->thre, h, t)[f].i, sigma = h->sigma_h;
      float wiener = POW2(r) + POW2(i);
      wiener = wiener / (wiener + (h->win) * POW2(sigma));
      row(h->coef, h, t)[f].r *= wiener;
      row(h->coef, h, t)[f].i *= wiener;
    }
}

static void overlap_add(BtOracle* h, int frames) { /* .c:284-300 and .c:630-641 */
  memcpy(h->outbuf, h->outbuf + h->macro, sizeof(float) * (size_t)h->half);
  memset(h->outbuf + h->half, 0, sizeof(float) * (size_t)h->macro);
  for (int i = 0; i < frames; i++) {
    kiss_fftr_inv(&h->fft, row(h->coef, h, i), h->inbuf_win);
    for (int j = 0; j < h->win; j++) h->outbuf[h->half * i + j] += h->inbuf_win[j] / (h->win);
  }
}

int bt_oracle_denoise_float(BtOracle* h, const float* in, int in_len) { /* .c:541-575 */
  if (in_len != h->half || !in) return MARS_ERROR_PARAMS;
  memcpy(h->inbuf, h->inbuf + h->half, sizeof(float) * (size_t)h->half);
  memcpy(h->inbuf + h->half, in, sizeof(float) * (size_t)h->half);
  for (int i = 0; i < h->win; i++) h->inbuf_win[i] = h->inbuf[i] * h->hann[i]; /* .c:273-282 */
  kiss_fftr_fwd(&h->fft, h->inbuf_win, row(h->coef, h, h->have));
  h->have++;
  if (h->have != NT) return MARS_NEED_MORE_SAMPLES;
  bt_core(h, NULL);
  overlap_add(h, NT);
  h->have = 0;
  return MARS_CAN_OUTPUT;
}

int bt_oracle_output_float(BtOracle* h, float* out, int out_len) { /* .c:589-601 */
  if (out_len < h->macro) return 0;
  memcpy(out, h->outbuf, sizeof(float) * (size_t)h->macro);
  return h->macro;
}

int bt_oracle_flush_float(BtOracle* h, float* out, int out_len) { /* .c:648-672 */
  const int out_size = h->have * h->half;
  if (out_len < out_size) return -1;
  overlap_add(h, h->have);
  memcpy(out, h->outbuf, sizeof(float) * (size_t)out_size);
  return out_size;
}

/* S16 conversions of the int16 entry points (.c:259-271) */
float bt_oracle_s16_to_float(int16_t v) {
  static const float kMaxInt16Inverse = 1.f / 32767;
  static const float kMinInt16Inverse = 1.f / (-32768);
  return v * (v > 0 ? kMaxInt16Inverse : -kMinInt16Inverse);
}
int16_t bt_oracle_float_to_s16(float v) {
  if (v > 0) return v >= 1 ? 32767 : (int16_t)(v * 32767 + 0.5f);
  return v <= -1 ? (-32768) : (int16_t)(-v * (-32768) - 0.5);
}

/* One whole macroblock (8 hops) of one stream; seg_out (optional) receives the
 * chosen (seg_time, seg_freq) of every macro-column. */
void bt_oracle_macroblock(BtOracle* h, const float* in, float* out, int* seg_out) {
  for (int t = 0; t < NT; ++t) {
    memcpy(h->inbuf, h->inbuf + h->half, sizeof(float) * (size_t)h->half);
    memcpy(h->inbuf + h->half, in + (size_t)t * h->half, sizeof(float) * (size_t)h->half);
    for (int i = 0; i < h->win; i++) h->inbuf_win[i] = h->inbuf[i] * h->hann[i];
    kiss_fftr_fwd(&h->fft, h->inbuf_win, row(h->coef, h, t));
  }
  bt_core(h, seg_out);
  overlap_add(h, NT);
  memcpy(out, h->outbuf, sizeof(float) * (size_t)h->macro);
  h->have = 0;
}

void bt_oracle_kiss_fftr(BtOracle* h, const float* timedata, float* freq) {
  kiss_fftr_fwd(&h->fft, timedata, (cpx*)freq);
}
void bt_oracle_kiss_fftri(BtOracle* h, const float* freq, float* timedata) {
  kiss_fftr_inv(&h->fft, (const cpx*)freq, timedata);
}
const float* bt_oracle_hann(const BtOracle* h) { return h->hann; }
