// ns_api.hip -- host side of include/asp_ns.h: table construction, the batch
// handle that owns the device state, state import/export, and the reference's
// per-stream WebRtcNs_* entry points implemented as a batch of one.
//
// There is NO CPU fallback: every entry point that computes needs a HIP device
// and fails with ASP_ERR_NO_DEVICE / -1 otherwise.
#include <hip/hip_runtime.h>

#include <chrono>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "asp_ns.h"
#include "ns_layout.h"

using namespace aspns;

namespace aspns {
hipError_t launch_ns_frame(int mode, float* state, int32_t* hist, const NsTables* T,
                           const float* in, float* out, int num_streams, hipStream_t s, bool g8 = false);
hipError_t launch_ns_frame1(bool io16, float* state, int32_t* hist, const NsTables* T,
                            const float* in, float* out, int num_streams, hipStream_t s,
                            unsigned long long* stamps = nullptr, int stamp_mode = 0);
hipError_t launch_ns_frame1_flow(bool io16, float* state, int32_t* hist, const NsTables* T,
                                 const float* in, float* out, int num_streams, hipStream_t s,
                                 unsigned* seq, unsigned* abort_w, unsigned want, int steps, int slot0, int ring,
                                 size_t per, unsigned long long* stamps = nullptr);
hipError_t launch_ns_frame2(bool io16, float* state, int32_t* hist, const NsTables* T,
                            const float* in, float* out, int num_streams, hipStream_t s,
                            unsigned long long* stamps = nullptr);
hipError_t launch_ns_frame2_flow(bool io16, float* state, int32_t* hist, const NsTables* T,
                                 const float* in, float* out, int num_streams, hipStream_t s,
                                 unsigned* seq, unsigned* abort_w, unsigned want, int steps, int slot0, int ring,
                                 size_t per, unsigned long long* stamps = nullptr);
hipError_t launch_ns_unpair(float* state, int num_streams, hipStream_t s);
hipError_t launch_ns_hb_live(const float* state, const NsTables* T, const float* in_low,
                             int32_t* live, int num_streams, int hist_off, hipStream_t s);
hipError_t launch_ns_hb_apply(const float* state, float* hb_tail, const int32_t* live,
                              const NsTables* T, const float* in_high, float* out_high,
                              int num_streams, int num_high, int paired, hipStream_t s);
hipError_t launch_ns_set_policy(float* state, int num_streams, int mode, float overdrive,
                                float denoiseBound, int gainmap, hipStream_t s);
hipError_t launch_rdft256(float* data, int count, int isgn, const NsTables* T, hipStream_t s, int n = 256);
hipError_t launch_debug_eval(int fn, float* data, size_t n, const NsTables* T, hipStream_t s);
hipError_t launch_debug_compare(int fn_a, int fn_b, unsigned start, unsigned count,
                                unsigned* n_bad, unsigned* bad_bits, float param,
                                const NsTables* T, hipStream_t s);
}  // namespace aspns

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess)
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(g_err, sizeof g_err, "%s", what);
  return code;
}

#define HIP_TRY(expr)                                         \
  do {                                                        \
    hipError_t e_ = (expr);                                   \
    if (e_ != hipSuccess) return fail(ASP_ERR_HIP, #expr, e_); \
  } while (0)

// ------------------------------------------------------------------ tables

unsigned bitrev(unsigned x, int bits) {
  unsigned r = 0;
  for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1u) << (bits - 1 - b);
  return r;
}

// kBlocks160w256 (ns/windows_private.h:94-147): sin(pi*i/192) ramps printed
// with 8 decimals and read back as (float)<double literal>.
float window_entry(int i) {
  if (i >= 96 && i <= 160) return 1.0f;
  char buf[32];
  double s = sin(M_PI * (double)(i < 96 ? i : 256 - i) / 192.0);
  snprintf(buf, sizeof buf, "%.8f", s);
  return (float)strtod(buf, NULL);
}

// kBlocks80w128 (ns/windows_private.h:64-91): sin(pi*i/96) over the 48-sample ramps, same text round trip.
float window8_entry(int i) {
  if (i >= 48 && i <= 80) return 1.0f;
  char buf[32];
  double s = sin(M_PI * (double)(i < 48 ? i : 128 - i) / 96.0);
  snprintf(buf, sizeof buf, "%.8f", s);
  return (float)strtod(buf, NULL);
}

// The 8 kHz geometry: tables of WebRtc_rdft(128) -- makewt(32), makect(32) (fft4g.c:642-690) -- as the
// per-lane twiddles of the 64-point transform of ns_kernels.hip (passes 1 and 2; pass 3 has none), the
// real-split factors, kBlocks80w128 and the start-up sums over bins 5..64.
void build_tables_8k(NsTables* T) {
  float w[32], c[32], tmp[32];
  const int nw = 32, nwh = 16;
  float delta = (float)atan((double)1.0f) / nwh;
  tmp[0] = 1;
  tmp[1] = 0;
  tmp[nwh] = (float)cos((double)(delta * nwh));
  tmp[nwh + 1] = tmp[nwh];
  for (int j = 2; j < nwh; j += 2) {
    float x = (float)cos((double)(delta * j));
    float y = (float)sin((double)(delta * j));
    tmp[j] = x;
    tmp[j + 1] = y;
    tmp[nw - j] = y;
    tmp[nw - j + 1] = x;
  }
  for (int j = 0; j < 16; ++j) {
    unsigned r = bitrev((unsigned)j, 4);
    w[2 * j] = tmp[2 * r];
    w[2 * j + 1] = tmp[2 * r + 1];
  }
  c[0] = (float)cos((double)(delta * nwh));
  c[nwh] = 0.5f * c[0];
  for (int j = 1; j < nwh; j++) {
    c[j] = 0.5f * (float)cos((double)(delta * j));
    c[nw - j] = 0.5f * (float)sin((double)(delta * j));
  }
  for (int i = 0; i < 128; ++i) T->window8[i] = window8_entry(i);
  for (int pass = 0; pass < 2; ++pass)
    for (int lane = 0; lane < 64; ++lane) {
      const int b = (lane >> 1) & 15, h = lane & 1;
      const int B = pass == 0 ? b : b >> 2;  // block index: same twiddle cases as the 128-point passes
      float tAr = 1.f, tAi = 0.f, tBr = 1.f, tBi = 0.f;
      bool diag = false;
      if (B == 1) {
        if (h == 0) {
          tBr = 0.f;
          tBi = 1.f;
        } else {
          diag = true;
          tAr = w[2];
        }
      } else if (B >= 2) {
        const int u = B >> 1;
        const float wk2r = w[2 * u], wk2i = w[2 * u + 1];
        float w1r, w1i, w3r, w3i, w2r, w2i;
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
        if (h == 0) {
          tBr = w2r;
          tBi = w2i;
        } else {
          tAr = w1r;
          tAi = w1i;
          tBr = w3r;
          tBi = w3i;
        }
      }
      T->tw8[pass][lane][0] = tAr;
      T->tw8[pass][lane][1] = tAi;
      T->tw8[pass][lane][2] = tBr;
      T->tw8[pass][lane][3] = tBi;
      if (diag) T->diag8[lane] |= 1 << pass;
    }
  for (int lane = 0; lane < 64; ++lane) {
    const int p = lane < 32 ? lane : 64 - lane;
    T->c8a[lane] = (lane == 0 || lane == 32) ? 0.f : c[p];
    T->c8b[lane] = (lane == 0 || lane == 32) ? 0.f : c[32 - p];
  }
  float sli = 0.f, slis = 0.f;  // ns_core.c:1094-1095 with magnLen 65, sequential over i = 5..64
  for (int i = 5; i < 65; ++i) {
    const float li = (float)log((double)(float)i);
    sli += li;
    slis += li * li;
  }
  T->sum_log_i8 = sli;
  T->sum_log_i_square8 = slis;
}

// NOTE: this file is C++, where cos(float) would resolve to the float overload;
// the reference is C, where cos() is the double function.  Every libm call
// below therefore casts its argument to double explicitly.
void build_tables(NsTables* T) {
  memset(T, 0, sizeof *T);
  float w[64], c[64], tmp[64];
  {  // makewt(64), utility/fft4g.c:642-669; bitrv2 == bit reversal of 32 complex entries
    const int nw = 64, nwh = 32;
    float delta = (float)atan((double)1.0f) / nwh;  // C semantics: double atan
    tmp[0] = 1;
    tmp[1] = 0;
    tmp[nwh] = (float)cos((double)(delta * nwh));
    tmp[nwh + 1] = tmp[nwh];
    for (int j = 2; j < nwh; j += 2) {
      float x = (float)cos((double)(delta * j));
      float y = (float)sin((double)(delta * j));
      tmp[j] = x;
      tmp[j + 1] = y;
      tmp[nw - j] = y;
      tmp[nw - j + 1] = x;
    }
    for (int j = 0; j < 32; ++j) {
      unsigned r = bitrev((unsigned)j, 5);
      w[2 * j] = tmp[2 * r];
      w[2 * j + 1] = tmp[2 * r + 1];
    }
  }
  {  // makect(64), utility/fft4g.c:671-690
    const int nc = 64, nch = 32;
    float delta = (float)atan((double)1.0f) / nch;
    c[0] = (float)cos((double)(delta * nch));
    c[nch] = 0.5f * c[0];
    for (int j = 1; j < nch; j++) {
      c[j] = 0.5f * (float)cos((double)(delta * j));
      c[nc - j] = 0.5f * (float)sin((double)(delta * j));
    }
  }
  for (int i = 0; i < kAnal; ++i) T->window[i] = window_entry(i);
  // Per-lane twiddles of the three radix-4 passes.  Lane = 2*b + h handles half
  // h of butterfly b; the block index B selects the reference's twiddle case
  // (fft4g.c:1008-1102 / 1114-1229): 0 none, 1 the w[2] block, 2u / 2u+1 general.
  for (int pass = 0; pass < 3; ++pass) {
    for (int lane = 0; lane < 64; ++lane) {
      const int b = lane >> 1, h = lane & 1;
      const int B = pass == 0 ? b : (pass == 1 ? b >> 2 : b >> 4);
      float tAr = 1.f, tAi = 0.f, tBr = 1.f, tBi = 0.f;
      bool diag = false;
      if (B == 1) {
        if (h == 0) {
          tBr = 0.f;
          tBi = 1.f;
        } else {
          // the reference's factored w[2] block: w (p.r - p.i), w (p.r + p.i) and -w (m.r + m.i),
          // w (m.r - m.i).  The pair-layout kernel forms p + i p and m - i m under the diag bit and
          // multiplies by (w, 0) / (-w, 0) (ns_pair_fft.h); ns_kernels.hip reads tAr only.
          diag = true;
          tAr = w[2];
          tBr = -w[2];
        }
      } else if (B >= 2) {
        const int u = B >> 1;
        const float wk2r = w[2 * u], wk2i = w[2 * u + 1];
        float w1r, w1i, w3r, w3i, w2r, w2i;
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
        if (h == 0) {
          tBr = w2r;
          tBi = w2i;
        } else {
          tAr = w1r;
          tAi = w1i;
          tBr = w3r;
          tBi = w3i;
        }
      }
      T->tw[pass][lane][0] = tAr;
      T->tw[pass][lane][1] = tAi;
      T->tw[pass][lane][2] = tBr;
      T->tw[pass][lane][3] = tBi;
      if (diag) T->diag[lane] |= 1 << pass;
    }
  }
  for (int q = 0; q < 64; ++q) {
    T->cq[q] = c[q];
    T->cr[q] = q == 0 ? 0.f : c[64 - q];
  }
  T->logi[0] = 0.f;
  for (int i = 1; i < kBins; ++i) T->logi[i] = (float)log((double)(float)i);
  float sli = 0.f, slis = 0.f;  // ns_core.c:1094-1095, sequential over i = 5..128
  for (int i = 5; i < kBins; ++i) {
    sli += T->logi[i];
    slis += T->logi[i] * T->logi[i];
  }
  T->sum_log_i = sli;
  T->sum_log_i_square = slis;
  for (int j = 0; j < 64; ++j) T->exp2_64[j] = exp2((double)j / 64.0);
  // log table: sub-interval i covers the float bit patterns [kLogOff + i 2^16, kLogOff + (i+1) 2^16);
  // c is its centre (exactly 1 for the interval around 1.0), invc = 1/c rounded to double and
  // log c = -log(invc) of that rounded value, both through long double
  for (int i = 0; i < 128; ++i) {
    const uint32_t b0 = 0x3f328000u + ((uint32_t)i << 16), b1 = b0 + (1u << 16);
    float f0, f1;
    memcpy(&f0, &b0, 4);
    memcpy(&f1, &b1, 4);
    long double c = ((long double)f0 + (long double)f1) / 2;
    if (f0 < 1.0f && f1 > 1.0f) c = 1.0L;
    const double invc = (double)(1.0L / c);
    T->logtab[i][0] = invc;
    T->logtab[i][1] = c == 1.0L ? 0.0 : (double)(-logl((long double)invc));
  }
  // full-butterfly twiddles of the two-streams-per-wave kernel (ns_kernels2.hip): block index B of pass 0/1/2 is
  // lane, lane >> 2, lane >> 4 (same twiddle cases as above, fft4g.c:1008-1102 / 1114-1229)
  for (int pass = 0; pass < 3; ++pass)
    for (int lane = 0; lane < 32; ++lane) {
      const int B = pass == 0 ? lane : (pass == 1 ? lane >> 2 : lane >> 4);
      float* e = T->tw2[pass][lane];
      e[0] = 1.f; e[1] = 0.f; e[2] = 1.f; e[3] = 0.f; e[4] = 1.f; e[5] = 0.f; e[6] = 0.f; e[7] = 0.f;
      if (B == 1) {
        e[0] = w[2];
        e[2] = 0.f;
        e[3] = 1.f;
        e[6] = 1.f;
      } else if (B >= 2) {
        const int u = B >> 1;
        const float wk2r = w[2 * u], wk2i = w[2 * u + 1];
        if ((B & 1) == 0) {
          const float w1r = w[4 * u], w1i = w[4 * u + 1];
          e[0] = w1r; e[1] = w1i; e[2] = wk2r; e[3] = wk2i;
          e[4] = w1r - 2 * wk2i * w1i;
          e[5] = 2 * wk2i * w1r - w1i;
        } else {
          const float w1r = w[4 * u + 2], w1i = w[4 * u + 3];
          e[0] = w1r; e[1] = w1i; e[2] = -wk2i; e[3] = wk2r;
          e[4] = w1r - 2 * wk2r * w1i;
          e[5] = 2 * wk2r * w1r - w1i;
        }
      }
    }
  for (int lane = 0; lane < 32; ++lane)
    for (int t = 0; t < 4; ++t) {
      const int g = lane >> 4, pq = (lane & 15) + 16 * t;  // element E = pq + 64 g
      float wkr = 0.f, wki = 0.f;
      if (pq != 0) {
        if (g == 0) {  // j side, j = pq (fft4g.c:1245-1246)
          wkr = 0.5f - c[64 - pq];
          wki = c[pq];
        } else {       // k side of pair j = 64 - pq
          wkr = 0.5f - c[pq];
          wki = c[64 - pq];
        }
      }
      T->spl[lane][t][0] = wkr;
      T->spl[lane][t][1] = wki;
    }
  build_tables_8k(T);
}

constexpr int kMaxDevices = 64;
std::mutex g_tab_mu;
NsTables* g_dev_tables[kMaxDevices] = {nullptr};

int device_tables(int device, NsTables** out) {
  if (device < 0 || device >= kMaxDevices) return fail(ASP_ERR_PARAM, "device ordinal out of range");
  std::lock_guard<std::mutex> lk(g_tab_mu);
  if (!g_dev_tables[device]) {
    NsTables* host = (NsTables*)malloc(sizeof(NsTables));
    build_tables(host);
    NsTables* dev = nullptr;
    hipError_t e = hipMalloc((void**)&dev, sizeof(NsTables));
    if (e == hipSuccess) e = hipMemcpy(dev, host, sizeof(NsTables), hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess) return fail(ASP_ERR_HIP, "uploading constant tables", e);
    g_dev_tables[device] = dev;
  }
  *out = g_dev_tables[device];
  return ASP_OK;
}

// The caller's current device is put back when an entry point returns (every entry point
// selects the batch's device for its HIP calls).
struct DeviceScope {
  int prev = -1;
  DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceScope() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

int select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(ASP_ERR_NO_DEVICE, "no HIP device available (the NS engine has no CPU fallback)", e);
  if (device < 0 || device >= n) return fail(ASP_ERR_PARAM, "device ordinal out of range");
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "hipSetDevice", e);
  return ASP_OK;
}

// --------------------------------------------------- canonical <-> device image

inline float i2f(int32_t v) {
  float f;
  memcpy(&f, &v, 4);
  return f;
}
inline int32_t f2i(float f) {
  int32_t v;
  memcpy(&v, &f, 4);
  return v;
}

// geometry of a stream (ns_core.c:89-98): 16 / 32 / 48 kHz 160 / 256 / 129, 8 kHz 80 / 128 / 65
inline int geo_block(int fs) { return fs == 8000 ? 80 : kBlockL; }
inline int geo_anal(int fs) { return fs == 8000 ? 128 : kAnal; }
inline int geo_bins(int fs) { return geo_anal(fs) / 2 + 1; }

// a per-bin array <-> its state row: bins 0 .. nb - 2 at their row positions, the last bin (128, or 64
// at 8 kHz) in the row's scalar slot
void vec_put(float* blk, int f, const float* src, int nb = kBins) {
  float* d = blk + kOffVec + f * kVecStride;
  for (int i = 128; i < kVecStride; ++i) d[i] = 0.f;
  for (int i = 0; i < nb - 1; ++i) blk[row_dword(f, i)] = src[i];
  blk[row_dword(f, 128)] = src[nb - 1];
}
void vec_get(const float* blk, int f, float* dst, int nb = kBins) {
  for (int i = 0; i < nb - 1; ++i) dst[i] = blk[row_dword(f, i)];
  dst[nb - 1] = blk[row_dword(f, 128)];
}

void pack_stream(const AspNsState* s, float* blk, int32_t* hist) {
  memset(blk, 0, sizeof(float) * kStreamDwords);
  float* sc = blk + kOffScalars;
  sc[S_BLOCKIND] = i2f(s->blockInd);
  sc[S_UPDATES] = i2f(s->updates);
  sc[S_COUNTER0] = i2f(s->counter[0]);
  sc[S_COUNTER1] = i2f(s->counter[1]);
  sc[S_COUNTER2] = i2f(s->counter[2]);
  for (int i = 0; i < 4; ++i) sc[S_MUP0 + i] = i2f(s->modelUpdatePars[i]);
  sc[S_GAINMAP] = i2f(s->gainmap);
  sc[S_AGGRMODE] = i2f(s->aggrMode);
  sc[S_INITFLAG] = i2f(s->initFlag);
  sc[S_OVERDRIVE] = s->overdrive;
  sc[S_DENOISEBOUND] = s->denoiseBound;
  sc[S_PRIORSPEECHPROB] = s->priorSpeechProb;
  sc[S_SIGNALENERGY] = s->signalEnergy;
  sc[S_SUMMAGN] = s->sumMagn;
  sc[S_WHITE] = s->whiteNoiseLevel;
  sc[S_PINKNUM] = s->pinkNoiseNumerator;
  sc[S_PINKEXP] = s->pinkNoiseExp;
  for (int i = 0; i < 7; ++i) sc[S_PMP0 + i] = s->priorModelPars[i];
  for (int i = 0; i < 7; ++i) sc[S_FD0 + i] = s->featureData[i];
  sc[S_FS] = i2f(s->fs);
  const int bl = geo_block(s->fs), carry = geo_anal(s->fs) - bl, nb = geo_bins(s->fs);
  memcpy(blk + kOffAnaHist, s->analyzeBuf + bl, sizeof(float) * carry);
  memcpy(blk + kOffDataHist, s->dataBuf + bl, sizeof(float) * carry);
  memcpy(blk + kOffSynt, s->syntBuf, sizeof(float) * carry);
  for (int k = 0; k < 3; ++k) {  // tracker k's bins start at k * magnLen (ns_core.c:224)
    vec_put(blk, V_LQ0 + k, s->lquantile + k * nb, nb);
    vec_put(blk, V_DEN0 + k, s->density + k * nb, nb);
  }
  vec_put(blk, V_QUANT, s->quantile, nb);
  vec_put(blk, V_SMOOTH, s->smooth, nb);
  vec_put(blk, V_NOISEPREV, s->noisePrev, nb);
  vec_put(blk, V_MAGNPREV_A, s->magnPrevAnalyze, nb);
  vec_put(blk, V_LOGLRT, s->logLrtTimeAvg, nb);
  vec_put(blk, V_AVGPAUSE, s->magnAvgPause, nb);
  vec_put(blk, V_NOISE, s->noise, nb);
  vec_put(blk, V_MAGNPREV_P, s->magnPrevProcess, nb);
  vec_put(blk, V_INITMAGN, s->initMagnEst, nb);
  vec_put(blk, V_PARAMNOISE, s->parametricNoise, nb);
  memset(hist, 0, sizeof(int32_t) * kHistDwords);
  memcpy(hist, s->histLrt, sizeof(int32_t) * kHist);
  memcpy(hist + kHistStride, s->histSpecFlat, sizeof(int32_t) * kHist);
  memcpy(hist + 2 * kHistStride, s->histSpecDiff, sizeof(int32_t) * kHist);
}

void unpack_stream(const float* blk, const int32_t* hist, bool paired, AspNsState* s) {
  memset(s, 0, sizeof *s);
  const float* sc = blk + kOffScalars;
  s->blockInd = f2i(sc[S_BLOCKIND]);
  s->updates = f2i(sc[S_UPDATES]);
  s->counter[0] = f2i(sc[S_COUNTER0]);
  s->counter[1] = f2i(sc[S_COUNTER1]);
  s->counter[2] = f2i(sc[S_COUNTER2]);
  for (int i = 0; i < 4; ++i) s->modelUpdatePars[i] = f2i(sc[S_MUP0 + i]);
  s->gainmap = f2i(sc[S_GAINMAP]);
  s->aggrMode = f2i(sc[S_AGGRMODE]);
  s->initFlag = f2i(sc[S_INITFLAG]);
  s->overdrive = sc[S_OVERDRIVE];
  s->denoiseBound = sc[S_DENOISEBOUND];
  s->priorSpeechProb = sc[S_PRIORSPEECHPROB];
  s->signalEnergy = sc[S_SIGNALENERGY];
  s->sumMagn = sc[S_SUMMAGN];
  s->whiteNoiseLevel = sc[S_WHITE];
  s->pinkNoiseNumerator = sc[S_PINKNUM];
  s->pinkNoiseExp = sc[S_PINKEXP];
  for (int i = 0; i < 7; ++i) s->priorModelPars[i] = sc[S_PMP0 + i];
  for (int i = 0; i < 7; ++i) s->featureData[i] = sc[S_FD0 + i];
  s->fs = f2i(sc[S_FS]);
  // only the live 96 (8 kHz: 48) samples of each sliding buffer exist on the device
  const int bl = geo_block(s->fs), carry = geo_anal(s->fs) - bl, nb = geo_bins(s->fs);
  memcpy(s->analyzeBuf + bl, blk + kOffAnaHist, sizeof(float) * carry);
  memcpy(s->dataBuf + bl, blk + (paired ? kOffAnaHist : kOffDataHist), sizeof(float) * carry);
  memcpy(s->syntBuf, blk + kOffSynt, sizeof(float) * carry);
  if (nb != kBins) {
    // the reference's Init fills these arrays to their full length (ns_core.c:116-131, :152-155) and
    // an 8 kHz stream never touches the part beyond its 3 x 65 / 65 bins
    for (int i = 0; i < ASP_NS_SIMULT * kBins; ++i) s->lquantile[i] = 8.f, s->density[i] = 0.3f;
    for (int i = 0; i < kBins; ++i) s->smooth[i] = 1.f, s->logLrtTimeAvg[i] = (float)0.5;
  }
  for (int k = 0; k < 3; ++k) {
    vec_get(blk, V_LQ0 + k, s->lquantile + k * nb, nb);
    vec_get(blk, V_DEN0 + k, s->density + k * nb, nb);
  }
  vec_get(blk, V_QUANT, s->quantile, nb);
  vec_get(blk, V_SMOOTH, s->smooth, nb);
  vec_get(blk, V_NOISEPREV, s->noisePrev, nb);
  vec_get(blk, V_MAGNPREV_A, s->magnPrevAnalyze, nb);
  vec_get(blk, V_LOGLRT, s->logLrtTimeAvg, nb);
  vec_get(blk, V_AVGPAUSE, s->magnAvgPause, nb);
  vec_get(blk, paired ? V_NOISEPREV : V_NOISE, s->noise, nb);
  vec_get(blk, paired ? V_MAGNPREV_A : V_MAGNPREV_P, s->magnPrevProcess, nb);
  vec_get(blk, V_INITMAGN, s->initMagnEst, nb);
  vec_get(blk, V_PARAMNOISE, s->parametricNoise, nb);
  memcpy(s->histLrt, hist, sizeof(int32_t) * kHist);
  memcpy(s->histSpecFlat, hist + kHistStride, sizeof(int32_t) * kHist);
  memcpy(s->histSpecDiff, hist + 2 * kHistStride, sizeof(int32_t) * kHist);
}

// WebRtcNs_InitCore (ns_core.c:74-214) as a canonical state.
void init_state(AspNsState* s, uint32_t fs) {
  memset(s, 0, sizeof *s);
  s->fs = (int32_t)fs;
  for (int i = 0; i < ASP_NS_SIMULT * kBins; i++) {
    s->lquantile[i] = 8.f;
    s->density[i] = 0.3f;
  }
  for (int i = 0; i < ASP_NS_SIMULT; i++)
    s->counter[i] = (int)floor((float)(200 * (i + 1)) / (float)ASP_NS_SIMULT);
  for (int i = 0; i < kBins; i++) s->smooth[i] = 1.f;
  s->priorSpeechProb = 0.5f;
  for (int i = 0; i < kBins; i++) s->logLrtTimeAvg[i] = (float)0.5;
  s->featureData[0] = (float)0.5;
  s->featureData[3] = (float)0.5;
  s->featureData[4] = (float)0.5;
  s->blockInd = -1;
  s->priorModelPars[0] = (float)0.5;
  s->priorModelPars[1] = 0.5f;
  s->priorModelPars[2] = 1.f;
  s->priorModelPars[3] = 0.5f;
  s->priorModelPars[4] = 1.f;
  s->modelUpdatePars[0] = 2;
  s->modelUpdatePars[1] = 500;
  s->modelUpdatePars[2] = 0;
  s->modelUpdatePars[3] = 500;
  s->overdrive = 1.f;  // WebRtcNs_set_policy_core(self, 0), ns_core.c:1020-1023
  s->denoiseBound = 0.5f;
  s->gainmap = 0;
  s->aggrMode = 0;
  s->initFlag = 1;
}

bool state_is_paired(const AspNsState* s) {
  const int bl = geo_block(s->fs), carry = geo_anal(s->fs) - bl;
  return memcmp(s->analyzeBuf + bl, s->dataBuf + bl, sizeof(float) * carry) == 0 &&
         memcmp(s->magnPrevAnalyze, s->magnPrevProcess, sizeof s->magnPrevAnalyze) == 0 &&
         memcmp(s->noise, s->noisePrev, sizeof s->noise) == 0;
}

}  // namespace

// ------------------------------------------------------------------- handle

struct AspNsBatch {
  int S = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  float* state = nullptr;
  int32_t* hist = nullptr;
  NsTables* tables = nullptr;
  float* stage_in = nullptr;   // device staging for ASP_MEM_HOST callers
  float* stage_out = nullptr;
  size_t stage_frames = 0;
  bool inited = false;
  bool paired = true;  // see ns_kernels.hip: fused step representation
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // Optional: the fused step of a large batch issued as `split` independent
  // sub-launches on separate HIP streams, so one part's load/store phases
  // overlap another part's arithmetic (streams never interact).
  int split = 1;
  // fused step kernel: 1 = ns_frame_kernel (one stream per wave, bins q / q + 64: ns_kernels.hip),
  // 3 = ns_frame1_kernel (one stream per wave, pair layout: ns_kernels1.hip)
  int kernel = 0;  // 0 = the default: 3 (measured faster than the others at every batch size, profiles/README.md)
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
  // A captured K-step replay (hipGraph): the launches of `g_steps` fused steps over the ring
  // (g_in, g_out, g_ring), still one launch per frame step and sub-launch; replayed while the
  // key is unchanged, so the host pays one graph launch instead of 2 K kernel launches.
  hipGraph_t graph[4] = {nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t gexec[4] = {nullptr, nullptr, nullptr, nullptr};
  int g_parts = 0;
  const float* g_in = nullptr;
  float* g_out = nullptr;
  int g_ring = 0, g_steps = 0, g_split = 0;
  bool g_io16 = false;
  int g_kernel = 0;
  bool use_graph = false;  // plain launches measured 4-7 % faster per step than graph replay (round 2)
  unsigned long long* timeline = nullptr;  // diagnostic (AspNsBatch_DebugTimeline): [workgroup][4] real-time stamps
  unsigned long long* flow_stamps = nullptr;  // diagnostic (AspNsBatch_DebugFlowStamps): 17 phase stamps of one wave
  double last_enqueue_us = 0.0;  // host time the last TimedSteps call spent enqueuing its launches
  // Hand-off build of the multi-step entry points (ns_kernels1.hip, NsFlowArgs): up to kFlowMaxSteps
  // consecutive frame steps of a K-step call per launch; a per-stream step counter in memory orders step
  // k + 1 of a stream behind its step k.  -1 = default (on for the pair-layout kernel), 0 = off, 1 = on.
  int flow = -1;
  unsigned* flow_seq = nullptr;    // [S] completed hand-off steps per stream (== flow_count between calls)
  unsigned* flow_abort = nullptr;  // 16 B: word 0 != 0 after a wait timed out
  unsigned flow_count = 0;         // hand-off steps enqueued so far
  bool flow_unchecked = false;     // hand-off launches enqueued since the abort word was last read
  // > 16 kHz: 1 or 2 high bands next to the low band (ns_core.c:1362-1414)
  uint32_t fs = 16000;
  int block = kBlockL;  // samples per frame and stream: 160, or 80 at 8 kHz (ns_core.c:89-98)
  int num_high = 0;
  float* hb_tail = nullptr;    // [S][2][96]: dataBufHB[b][160..255]
  int32_t* hb_live = nullptr;  // [S]: energy1 != 0 of the frame being processed
  float* hb_stage = nullptr;   // host-memory callers: [2][num_high][S][160]
};

namespace {

int ensure_stage(AspNsBatch* b, size_t frames) {
  if (b->stage_frames >= frames) return ASP_OK;
  if (b->stage_in) (void)hipFree(b->stage_in);
  if (b->stage_out) (void)hipFree(b->stage_out);
  b->stage_in = b->stage_out = nullptr;
  b->stage_frames = 0;
  const size_t bytes = frames * (size_t)b->S * b->block * sizeof(float);
  HIP_TRY(hipMalloc((void**)&b->stage_in, bytes));
  HIP_TRY(hipMalloc((void**)&b->stage_out, bytes));
  b->stage_frames = frames;
  return ASP_OK;
}

int ensure_unpaired(AspNsBatch* b) {
  if (!b->paired) return ASP_OK;
  HIP_TRY(launch_ns_unpair(b->state, b->S, b->stream));
  b->paired = false;
  return ASP_OK;
}

int check(AspNsBatch* b) {
  if (!b) return fail(ASP_ERR_PARAM, "null batch handle");
  if (!b->inited) return fail(ASP_ERR_STATE, "batch not initialised (call AspNsBatch_Init)");
  hipError_t e = hipSetDevice(b->device);
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "hipSetDevice", e);
  return ASP_OK;
}

}  // namespace

extern "C" {

const char* AspNs_last_error(void) { return g_err; }

// Host-side copy of the constant tables (window, per-lane twiddles, ...): no
// device needed, so the CPU test-suite can pin them against the oracle.
int AspNs_host_tables(void* out, size_t bytes) {
  if (!out || bytes != sizeof(NsTables)) return fail(ASP_ERR_PARAM, "AspNs_host_tables: size mismatch");
  build_tables((NsTables*)out);
  return ASP_OK;
}
size_t AspNs_host_tables_size(void) { return sizeof(NsTables); }

int AspNs_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(ASP_ERR_NO_DEVICE, "hipGetDeviceCount", e);
  return n;
}

int AspNsBatch_Create(AspNsBatch** out, int num_streams, int device) {
  if (!out || num_streams <= 0) return fail(ASP_ERR_PARAM, "AspNsBatch_Create: bad argument");
  *out = nullptr;
  DeviceScope dev_scope_;
  int rc = select_device(device);
  if (rc) return rc;
  AspNsBatch* b = new AspNsBatch();
  b->S = num_streams;
  b->device = device;
  rc = device_tables(device, &b->tables);
  if (rc) {
    delete b;
    return rc;
  }
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e == hipSuccess) b->own_stream = true;
  if (e == hipSuccess) e = hipMalloc((void**)&b->state, (size_t)num_streams * kStreamDwords * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&b->hist, (size_t)num_streams * kHistDwords * 4);
  if (e == hipSuccess) e = hipEventCreate(&b->ev0);
  if (e == hipSuccess) e = hipEventCreate(&b->ev1);
  if (e != hipSuccess) {
    AspNsBatch_Free(b);
    return fail(ASP_ERR_HIP, "AspNsBatch_Create: device allocation", e);
  }
  *out = b;
  return ASP_OK;
}

int AspNsBatch_Free(AspNsBatch* b) {
  if (!b) return ASP_OK;
  DeviceScope dev_scope_;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->state) (void)hipFree(b->state);
  if (b->hist) (void)hipFree(b->hist);
  if (b->flow_seq) (void)hipFree(b->flow_seq);
  if (b->flow_abort) (void)hipFree(b->flow_abort);
  if (b->stage_in) (void)hipFree(b->stage_in);
  if (b->stage_out) (void)hipFree(b->stage_out);
  if (b->hb_tail) (void)hipFree(b->hb_tail);
  if (b->hb_live) (void)hipFree(b->hb_live);
  if (b->hb_stage) (void)hipFree(b->hb_stage);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  for (int p = 0; p < 4; ++p) {
    if (b->gexec[p]) (void)hipGraphExecDestroy(b->gexec[p]);
    if (b->graph[p]) (void)hipGraphDestroy(b->graph[p]);
  }
  if (b->fork_ev) (void)hipEventDestroy(b->fork_ev);
  for (int i = 0; i < 3; ++i) {
    if (b->join_ev[i]) (void)hipEventDestroy(b->join_ev[i]);
    if (b->side[i]) (void)hipStreamDestroy(b->side[i]);
  }
  if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return ASP_OK;
}

int AspNsBatch_num_streams(const AspNsBatch* b) { return b ? b->S : ASP_ERR_PARAM; }

int AspNsBatch_Init(AspNsBatch* b, uint32_t fs) {
  if (!b) return fail(ASP_ERR_PARAM, "null batch handle");
  // ns_core.c:82-86: 8 kHz (80 / 128 / 65: ns_kernels.hip's G8 instantiation serves every entry point)
  // and 16 / 32 / 48 kHz (160 / 256 / 129)
  if (fs != 8000 && fs != 16000 && fs != 32000 && fs != 48000)
    return fail(ASP_ERR_PARAM, "AspNsBatch_Init: fs must be 8000, 16000, 32000 or 48000");
  HIP_TRY(hipSetDevice(b->device));
  if (b->fs != fs && b->stage_frames) {  // the staging buffers are sized in frames of the old length
    if (b->stream) HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->stage_in) (void)hipFree(b->stage_in);
    if (b->stage_out) (void)hipFree(b->stage_out);
    b->stage_in = b->stage_out = nullptr;
    b->stage_frames = 0;
  }
  b->fs = fs;
  b->block = geo_block((int)fs);
  b->num_high = fs >= 32000 ? (int)(fs / 16000) - 1 : 0;
  if (b->num_high > 0) {
    if (!b->hb_tail) HIP_TRY(hipMalloc((void**)&b->hb_tail, (size_t)b->S * 2 * kCarry * sizeof(float)));
    if (!b->hb_live) HIP_TRY(hipMalloc((void**)&b->hb_live, (size_t)b->S * sizeof(int32_t)));
    if (!b->hb_stage)
      HIP_TRY(hipMalloc((void**)&b->hb_stage, (size_t)2 * 2 * b->S * kBlockL * sizeof(float)));
    // on the batch's own (non-blocking) stream: a null-stream memset is not ordered with its kernels
    HIP_TRY(hipMemsetAsync(b->hb_tail, 0, (size_t)b->S * 2 * kCarry * sizeof(float), b->stream));  // ns_core.c:110-112
  }
  AspNsState* s0 = (AspNsState*)malloc(sizeof(AspNsState));
  init_state(s0, fs);
  const int chunk = b->S < 256 ? b->S : 256;
  std::vector<float> blk((size_t)chunk * kStreamDwords);
  std::vector<int32_t> hh(kHistDwords);
  pack_stream(s0, blk.data(), hh.data());
  free(s0);
  for (int i = 1; i < chunk; ++i)
    memcpy(blk.data() + (size_t)i * kStreamDwords, blk.data(), sizeof(float) * kStreamDwords);
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (int s = 0; s < b->S; s += chunk) {
    const int n = (b->S - s) < chunk ? (b->S - s) : chunk;
    HIP_TRY(hipMemcpy(b->state + (size_t)s * kStreamDwords, blk.data(),
                      (size_t)n * kStreamDwords * 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemsetAsync(b->hist, 0, (size_t)b->S * kHistDwords * 4, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->inited = true;
  b->paired = true;
  return ASP_OK;
}

static int flow_check(AspNsBatch* b);

// WebRtcNs_Init of ONE stream of a running batch (noise_suppression.c:35-39 per handle): its state block and
// histograms go back to InitCore's values at the batch's sample rate (policy 0, as InitCore ends,
// ns_core.c:207); the other streams are untouched.  Ordered on the batch's stream like every other call.
int AspNsBatch_InitStream(AspNsBatch* b, int stream) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (stream < 0 || stream >= b->S) return fail(ASP_ERR_PARAM, "InitStream: stream out of range");
  std::vector<AspNsState> s0(1);
  init_state(s0.data(), b->fs);
  std::vector<float> blk(kStreamDwords);
  std::vector<int32_t> hh(kHistDwords);
  pack_stream(s0.data(), blk.data(), hh.data());  // every row, hot and cold: valid in the fused and the two-call representation
  HIP_TRY(hipStreamSynchronize(b->stream));
  rc = flow_check(b);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(b->state + (size_t)stream * kStreamDwords, blk.data(), kStreamDwords * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->hist + (size_t)stream * kHistDwords, hh.data(), kHistDwords * 4, hipMemcpyHostToDevice));
  if (b->num_high > 0)  // ns_core.c:110-112: the high bands' delay lines
    HIP_TRY(hipMemset(b->hb_tail + (size_t)stream * 2 * kCarry, 0, (size_t)2 * kCarry * sizeof(float)));
  return ASP_OK;
}

static const float kPolicyOver[4] = {1.f, 1.f, 1.1f, 1.25f};    // ns_core.c:1020-1039
static const float kPolicyBound[4] = {0.5f, 0.25f, 0.125f, 0.09f};
static const int kPolicyMap[4] = {0, 1, 1, 1};

// WebRtcNs_set_policy of ONE stream (noise_suppression.c:41-44 per handle).
int AspNsBatch_set_policy_stream(AspNsBatch* b, int stream, int mode) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (stream < 0 || stream >= b->S) return fail(ASP_ERR_PARAM, "set_policy_stream: stream out of range");
  if (mode < 0 || mode > 3) return fail(ASP_ERR_PARAM, "set_policy: mode must be 0..3");
  HIP_TRY(launch_ns_set_policy(b->state + (size_t)stream * kStreamDwords, 1, mode, kPolicyOver[mode], kPolicyBound[mode],
                               kPolicyMap[mode], b->stream));
  return ASP_OK;
}

int AspNsBatch_set_policy(AspNsBatch* b, int mode) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (mode < 0 || mode > 3) return fail(ASP_ERR_PARAM, "set_policy: mode must be 0..3");
  // ns_core.c:1020-1039
  static const float kOver[4] = {1.f, 1.f, 1.1f, 1.25f};
  static const float kBound[4] = {0.5f, 0.25f, 0.125f, 0.09f};
  static const int kMap[4] = {0, 1, 1, 1};
  HIP_TRY(launch_ns_set_policy(b->state, b->S, mode, kOver[mode], kBound[mode], kMap[mode],
                               b->stream));
  return ASP_OK;
}

// One fused paired frame step over streams [s0, s0 + n) of the batch.
static hipError_t fused_launch(AspNsBatch* b, bool io16, const float* din, float* dout, int s0, int n,
                               hipStream_t st) {
  const size_t sper = io16 ? b->block / 2 : b->block;  // one stream's frame in float units
  float* state = b->state + (size_t)s0 * kStreamDwords;
  int32_t* hist = b->hist + (size_t)s0 * kHistDwords;
  const float* in = din + (size_t)s0 * sper;
  float* out = dout + (size_t)s0 * sper;
  if (b->kernel == 1 || b->fs == 8000)  // the pair-layout kernel is built for the 129-bin geometry only
    return launch_ns_frame(io16 ? 3 : 2, state, hist, b->tables, in, out, n, st, b->fs == 8000);
  if (b->kernel == 2)  // two streams per wave (ns_kernels2.hip); an odd count leaves the last wave's high half idle
    return launch_ns_frame2(io16, state, hist, b->tables, in, out, n, st);
  // timeline diagnostic: this sub-launch's first workgroup's slot, stamp mode 1
  unsigned long long* tl = b->timeline ? b->timeline + (size_t)(s0 / 4) * 4 : nullptr;
  return launch_ns_frame1(io16, state, hist, b->tables, in, out, n, st, tl, tl ? 1 : 0);
}

// stream boundaries of the sub-launch chains: multiples of 8 streams (two workgroups of 4)
static int chain_parts(const AspNsBatch* b, int base[5]) {
  const int parts = (b->split > 1 && b->S >= 8 * b->split) ? b->split : 1;
  for (int p = 0; p <= parts; ++p) base[p] = (int)(((long long)b->S * p / parts) / 8 * 8);
  base[parts] = b->S;
  return parts;
}

static bool flow_default() {
  const char* e = getenv("ASP_NS_FLOW");
  return !(e && e[0] == '0');
}

// The hand-off build serves a K-step call of a batch in the fused representation on the pair-layout kernel.
static bool flow_applies(const AspNsBatch* b, int steps) {
  const bool on = b->flow < 0 ? flow_default() : b->flow != 0;
  // (the two-streams-per-wave kernel addresses the whole state array through one 32-bit buffer offset)
  if (b->kernel == 2 && (size_t)b->S * kStreamDwords * 4 >= ((size_t)1 << 32)) return false;
  return on && steps >= 2 && b->paired && b->kernel != 1 && b->fs != 8000 && b->timeline == nullptr;
}

static int flow_resources(AspNsBatch* b) {
  if (!b->flow_seq) {
    HIP_TRY(hipMalloc((void**)&b->flow_seq, (size_t)b->S * sizeof(unsigned)));
    HIP_TRY(hipMalloc((void**)&b->flow_abort, 16));
    HIP_TRY(hipMemsetAsync(b->flow_seq, 0, (size_t)b->S * sizeof(unsigned), b->stream));
    HIP_TRY(hipMemsetAsync(b->flow_abort, 0, 16, b->stream));
    b->flow_count = 0;
  }
  return ASP_OK;
}

// K steps of the hand-off build on the batch's stream: launches of up to kFlowMaxSteps consecutive frame
// steps each (grid y = step); launches follow each other in stream order.
constexpr int kFlowMaxSteps = 64;
static int flow_steps(AspNsBatch* b, const float* din, float* dout, int ring, int steps, bool io16) {
  int rc = flow_resources(b);
  if (rc) return rc;
  const size_t per = (size_t)b->S * b->block / (io16 ? 2 : 1);
  int maxm = kFlowMaxSteps;
  if (const char* e = getenv("ASP_NS_FLOW_MAX")) {  // tuning: frame steps per launch
    const int v = atoi(e);
    if (v >= 2 && v <= kFlowMaxSteps) maxm = v;
  }
  for (int k = 0; k < steps; k += maxm) {
    const int m = steps - k < maxm ? steps - k : maxm;
    if (b->kernel == 2)
      HIP_TRY(launch_ns_frame2_flow(io16, b->state, b->hist, b->tables, din, dout, b->S, b->stream, b->flow_seq,
                                    b->flow_abort, b->flow_count, m, k % ring, ring, per, b->flow_stamps));
    else
      HIP_TRY(launch_ns_frame1_flow(io16, b->state, b->hist, b->tables, din, dout, b->S, b->stream, b->flow_seq,
                                    b->flow_abort, b->flow_count, m, k % ring, ring, per, b->flow_stamps));
    b->flow_count += (unsigned)m;
  }
  b->flow_unchecked = true;
  return ASP_OK;
}

// After the batch's stream has been synchronised: did a hand-off wait time out?  (It cannot while the
// launches of one batch run as enqueued; a timeout means steps were skipped, so the call fails loudly and
// the counters are put back in step.)
static int flow_check(AspNsBatch* b) {
  if (!b->flow_unchecked) return ASP_OK;
  b->flow_unchecked = false;
  unsigned a = 0;
  HIP_TRY(hipMemcpy(&a, b->flow_abort, sizeof a, hipMemcpyDeviceToHost));
  if (a == 0) return ASP_OK;
  std::vector<unsigned> seq((size_t)b->S, b->flow_count);
  HIP_TRY(hipMemcpy(b->flow_seq, seq.data(), seq.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(b->flow_abort, 0, 16));
  return fail(ASP_ERR_HIP, "NS hand-off wait timed out: frame steps were skipped, re-initialise the batch");
}

// `steps` fused frame steps on device buffers; step k reads/writes ring slot k % ring.
static int fused_steps(AspNsBatch* b, const float* din, float* dout, int ring, int steps,
                       bool io16 = false) {
  if (flow_applies(b, steps)) return flow_steps(b, din, dout, ring, steps, io16);
  // offsets below are in float units; int16 frames are half as wide
  const size_t per = (size_t)b->S * b->block / (io16 ? 2 : 1);

  int base[5];
  const int parts = chain_parts(b, base);
  if (parts == 1) {
    for (int k = 0; k < steps; ++k) {
      const size_t off = per * (size_t)(k % ring);
      HIP_TRY(fused_launch(b, io16, din + off, dout + off, 0, b->S, b->stream));
    }
    return ASP_OK;
  }
  HIP_TRY(hipEventRecord(b->fork_ev, b->stream));
  for (int p = 1; p < parts; ++p) HIP_TRY(hipStreamWaitEvent(b->side[p - 1], b->fork_ev, 0));
  for (int k = 0; k < steps; ++k) {
    const size_t off = per * (size_t)(k % ring);
    for (int p = 0; p < parts; ++p) {
      hipStream_t st = p == 0 ? b->stream : b->side[p - 1];
      const int s0 = base[p], n = base[p + 1] - base[p];
      HIP_TRY(fused_launch(b, io16, din + off, dout + off, s0, n, st));
    }
  }
  for (int p = 1; p < parts; ++p) {
    HIP_TRY(hipEventRecord(b->join_ev[p - 1], b->side[p - 1]));
    HIP_TRY(hipStreamWaitEvent(b->stream, b->join_ev[p - 1], 0));
  }
  return ASP_OK;
}

static void drop_graph(AspNsBatch* b) {
  for (int p = 0; p < 4; ++p) {
    if (b->gexec[p]) (void)hipGraphExecDestroy(b->gexec[p]);
    if (b->graph[p]) (void)hipGraphDestroy(b->graph[p]);
    b->gexec[p] = nullptr;
    b->graph[p] = nullptr;
  }
  b->g_steps = 0;
}

// The same launches as fused_steps(), captured as ONE LINEAR hipGraph PER CHAIN (kernel nodes only,
// one per frame step; a chain's graph is replayed on the chain's own HIP stream, so a replay is the
// chain's in-order sequence of launches without a host enqueue per launch).  One graph with the chains
// as parallel branches measured 15-20 % slower per step than plain launches (round 2), linear graphs
// do not.  The captures are kept while (buffers, ring, steps, split, kernel) stay the same.
static int ensure_graph(AspNsBatch* b, const float* din, float* dout, int ring, int steps, bool io16) {
  const bool hit = b->gexec[0] && b->g_in == din && b->g_out == dout && b->g_ring == ring &&
                   b->g_steps == steps && b->g_split == b->split && b->g_io16 == io16 &&
                   b->g_kernel == b->kernel;
  if (hit) return ASP_OK;
  drop_graph(b);
  int base[5];
  const int parts = chain_parts(b, base);
  const size_t per = (size_t)b->S * b->block / (io16 ? 2 : 1);
  for (int p = 0; p < parts; ++p) {
    hipStream_t st = p == 0 ? b->stream : b->side[p - 1];
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipError_t le = hipSuccess;
    for (int k = 0; k < steps && le == hipSuccess; ++k) {
      const size_t off = per * (size_t)(k % ring);
      le = fused_launch(b, io16, din + off, dout + off, base[p], base[p + 1] - base[p], st);
    }
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (le != hipSuccess || e != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      drop_graph(b);
      return fail(ASP_ERR_HIP, "graph capture of the frame steps", le != hipSuccess ? le : e);
    }
    b->graph[p] = g;
    e = hipGraphInstantiate(&b->gexec[p], g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
      drop_graph(b);
      return fail(ASP_ERR_HIP, "hipGraphInstantiate", e);
    }
  }
  b->g_parts = parts;
  b->g_in = din;
  b->g_out = dout;
  b->g_ring = ring;
  b->g_steps = steps;
  b->g_split = b->split;
  b->g_io16 = io16;
  b->g_kernel = b->kernel;
  return ASP_OK;
}

// replay: fork the side streams off the batch's stream, one graph launch per chain, join
static int launch_graphs(AspNsBatch* b) {
  const int parts = b->g_parts;
  if (parts > 1) {
    HIP_TRY(hipEventRecord(b->fork_ev, b->stream));
    for (int p = 1; p < parts; ++p) HIP_TRY(hipStreamWaitEvent(b->side[p - 1], b->fork_ev, 0));
  }
  for (int p = 0; p < parts; ++p) HIP_TRY(hipGraphLaunch(b->gexec[p], p == 0 ? b->stream : b->side[p - 1]));
  for (int p = 1; p < parts; ++p) {
    HIP_TRY(hipEventRecord(b->join_ev[p - 1], b->side[p - 1]));
    HIP_TRY(hipStreamWaitEvent(b->stream, b->join_ev[p - 1], 0));
  }
  return ASP_OK;
}

static int run_frames(AspNsBatch* b, int kmode, const float* in, float* out, int num_frames,
                      int mem) {
  const size_t per = (size_t)b->S * b->block;
  const bool g8 = b->fs == 8000;
  const float* din = in;
  float* dout = out;
  if (mem == ASP_MEM_HOST) {
    int rc = ensure_stage(b, (size_t)num_frames);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->stage_in, in, per * num_frames * 4, hipMemcpyHostToDevice,
                           b->stream));
    din = b->stage_in;
    dout = b->stage_out;
  } else if (mem != ASP_MEM_DEVICE) {
    return fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  if (kmode == 2 && b->paired && num_frames > 0) {
    int rc = fused_steps(b, din, dout, num_frames, num_frames);
    if (rc) return rc;
  } else
  for (int f = 0; f < num_frames; ++f) {
    const float* fi = din + per * f;
    float* fo = dout ? dout + per * f : nullptr;
    if (kmode == 2 && !b->paired) {
      HIP_TRY(launch_ns_frame(0, b->state, b->hist, b->tables, fi, fo, b->S, b->stream, g8));
      HIP_TRY(launch_ns_frame(1, b->state, b->hist, b->tables, fi, fo, b->S, b->stream, g8));
    } else {
      HIP_TRY(launch_ns_frame(kmode, b->state, b->hist, b->tables, fi, fo, b->S, b->stream, g8));
    }
  }
  if (mem == ASP_MEM_HOST) {
    if (out && kmode != 0)
      HIP_TRY(hipMemcpyAsync(out, b->stage_out, per * num_frames * 4, hipMemcpyDeviceToHost,
                             b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return flow_check(b);
  }
  return ASP_OK;
}

int AspNsBatch_Analyze(AspNsBatch* b, const float* frames, int mem) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!frames) return fail(ASP_ERR_PARAM, "Analyze: null frames");
  rc = ensure_unpaired(b);
  if (rc) return rc;
  return run_frames(b, 0, frames, nullptr, 1, mem);
}

int AspNsBatch_Process(AspNsBatch* b, const float* in, float* out, int mem) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in || !out) return fail(ASP_ERR_PARAM, "Process: null frames");
  rc = ensure_unpaired(b);
  if (rc) return rc;
  return run_frames(b, 1, in, out, 1, mem);
}

int AspNsBatch_AnalyzeProcess(AspNsBatch* b, const float* in, float* out, int num_frames,
                              int mem) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in || !out || num_frames < 0) return fail(ASP_ERR_PARAM, "AnalyzeProcess: bad argument");
  if (b->num_high > 0)  // the high-band delay line would fall out of step (ns_core.c:1227-1235)
    return fail(ASP_ERR_STATE, "AnalyzeProcess: one-band entry point on a batch initialised at 32 / 48 kHz (use AnalyzeProcessBands)");
  return run_frames(b, 2, in, out, num_frames, mem);
}

// One frame of every stream with its high band(s): `fused` = Analyze + Process of the low band in
// one step (paired state), otherwise Process alone (the caller ran Analyze).  Device pointers.
static int bands_frame_device(AspNsBatch* b, bool fused, const float* low_in, const float* high_in,
                              float* low_out, float* high_out) {
  const int hist_off = b->paired ? kOffAnaHist : kOffDataHist;
  HIP_TRY(launch_ns_hb_live(b->state, b->tables, low_in, b->hb_live, b->S, hist_off, b->stream));
  if (fused && b->paired) {
    HIP_TRY(fused_launch(b, false, low_in, low_out, 0, b->S, b->stream));
  } else {
    if (fused)  // unpaired streams: Analyze then Process as two launches
      HIP_TRY(launch_ns_frame(0, b->state, b->hist, b->tables, low_in, low_out, b->S, b->stream));
    HIP_TRY(launch_ns_frame(1, b->state, b->hist, b->tables, low_in, low_out, b->S, b->stream));
  }
  HIP_TRY(launch_ns_hb_apply(b->state, b->hb_tail, b->hb_live, b->tables, high_in, high_out, b->S,
                             b->num_high, b->paired ? 1 : 0, b->stream));
  return ASP_OK;
}

static int bands_run(AspNsBatch* b, bool fused, const float* low_in, const float* high_in,
                     float* low_out, float* high_out, int num_frames, int mem) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (b->num_high < 1) return fail(ASP_ERR_STATE, "bands: the batch was initialised at 16 kHz (one band)");
  if (!low_in || !high_in || !low_out || !high_out || num_frames < 0)
    return fail(ASP_ERR_PARAM, "bands: bad argument");
  const size_t lper = (size_t)b->S * kBlockL, hper = lper * b->num_high;
  if (mem == ASP_MEM_DEVICE) {
    for (int f = 0; f < num_frames; ++f) {
      rc = bands_frame_device(b, fused, low_in + lper * f, high_in + hper * f, low_out + lper * f,
                              high_out + hper * f);
      if (rc) return rc;
    }
    return ASP_OK;
  }
  if (mem != ASP_MEM_HOST) return fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  rc = ensure_stage(b, 1);
  if (rc) return rc;
  float* hin = b->hb_stage;
  float* hout = b->hb_stage + (size_t)2 * lper;
  for (int f = 0; f < num_frames; ++f) {
    HIP_TRY(hipMemcpyAsync(b->stage_in, low_in + lper * f, lper * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(hin, high_in + hper * f, hper * 4, hipMemcpyHostToDevice, b->stream));
    rc = bands_frame_device(b, fused, b->stage_in, hin, b->stage_out, hout);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(low_out + lper * f, b->stage_out, lper * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(high_out + hper * f, hout, hper * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return ASP_OK;
}

int AspNsBatch_AnalyzeProcessBands(AspNsBatch* b, const float* low_in, const float* high_in,
                                   float* low_out, float* high_out, int num_frames, int mem) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  return bands_run(b, true, low_in, high_in, low_out, high_out, num_frames, mem);
}

int AspNsBatch_ProcessBands(AspNsBatch* b, const float* low_in, const float* high_in, float* low_out,
                            float* high_out, int mem) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  rc = ensure_unpaired(b);
  if (rc) return rc;
  return bands_run(b, false, low_in, high_in, low_out, high_out, 1, mem);
}

int AspNsBatch_num_bands(const AspNsBatch* b) { return b ? 1 + b->num_high : ASP_ERR_PARAM; }

int AspNsBatch_ExportHbState(AspNsBatch* b, int stream, AspNsHbState* out) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!out || stream < 0 || stream >= b->S || b->num_high < 1)
    return fail(ASP_ERR_PARAM, "ExportHbState: bad argument");
  HIP_TRY(hipStreamSynchronize(b->stream));
  float tail[2 * kCarry];
  HIP_TRY(hipMemcpy(tail, b->hb_tail + (size_t)stream * 2 * kCarry, sizeof tail, hipMemcpyDeviceToHost));
  memset(out, 0, sizeof *out);
  for (int k = 0; k < 2; ++k) memcpy(out->dataBufHB[k] + kBlockL, tail + k * kCarry, kCarry * sizeof(float));
  return ASP_OK;
}

int AspNsBatch_ImportHbState(AspNsBatch* b, int stream, const AspNsHbState* in) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in || stream < 0 || stream >= b->S || b->num_high < 1)
    return fail(ASP_ERR_PARAM, "ImportHbState: bad argument");
  HIP_TRY(hipStreamSynchronize(b->stream));
  float tail[2 * kCarry];
  for (int k = 0; k < 2; ++k) memcpy(tail + k * kCarry, in->dataBufHB[k] + kBlockL, kCarry * sizeof(float));
  HIP_TRY(hipMemcpy(b->hb_tail + (size_t)stream * 2 * kCarry, tail, sizeof tail, hipMemcpyHostToDevice));
  return ASP_OK;
}

int AspNsBatch_AnalyzeProcessS16(AspNsBatch* b, const int16_t* in, int16_t* out, int num_frames,
                                 int mem) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in || !out || num_frames < 0) return fail(ASP_ERR_PARAM, "AnalyzeProcessS16: bad argument");
  if (!b->paired)
    return fail(ASP_ERR_STATE, "AnalyzeProcessS16 needs streams driven only through the fused step");
  if (b->num_high > 0)
    return fail(ASP_ERR_STATE, "AnalyzeProcessS16: one-band entry point on a batch initialised at 32 / 48 kHz");
  if (num_frames == 0) return ASP_OK;
  const size_t bytes = (size_t)b->S * b->block * sizeof(int16_t) * (size_t)num_frames;
  const float* din = reinterpret_cast<const float*>(in);
  float* dout = reinterpret_cast<float*>(out);
  if (mem == ASP_MEM_HOST) {
    rc = ensure_stage(b, (size_t)(num_frames + 1) / 2);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->stage_in, in, bytes, hipMemcpyHostToDevice, b->stream));
    din = b->stage_in;
    dout = b->stage_out;
  } else if (mem != ASP_MEM_DEVICE) {
    return fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  rc = fused_steps(b, din, dout, num_frames, num_frames, true);
  if (rc) return rc;
  if (mem == ASP_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(out, b->stage_out, bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return flow_check(b);
  }
  return ASP_OK;
}

int AspNsBatch_AnalyzeProcessReplay(AspNsBatch* b, const float* in, float* out, int frames_in_ring,
                                    int steps) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in || !out || frames_in_ring <= 0 || steps < 0)
    return fail(ASP_ERR_PARAM, "AnalyzeProcessReplay: bad argument");
  if (!b->paired) return fail(ASP_ERR_STATE, "AnalyzeProcessReplay needs the fused (paired) representation");
  if (b->num_high > 0) return fail(ASP_ERR_STATE, "AnalyzeProcessReplay: one-band entry point on a multi-band batch");
  if (steps == 0) return ASP_OK;
  if (!b->use_graph || b->stream == nullptr) return fused_steps(b, in, out, frames_in_ring, steps);
  rc = ensure_graph(b, in, out, frames_in_ring, steps, false);
  if (rc) return rc;
  return launch_graphs(b);
}

int AspNsBatch_TimedSteps(AspNsBatch* b, const float* in, float* out, int frames_in_ring,
                          int steps, float* elapsed_ms) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in || !out || frames_in_ring <= 0 || steps < 0 || !elapsed_ms)
    return fail(ASP_ERR_PARAM, "TimedSteps: bad argument");
  if (!b->paired) return fail(ASP_ERR_STATE, "TimedSteps needs the fused (paired) representation");
  if (b->num_high > 0) return fail(ASP_ERR_STATE, "TimedSteps: one-band entry point on a batch initialised at 32 / 48 kHz");
  // graph replay: capture + instantiate outside the timed region (the legacy null stream cannot capture)
  const bool graph = b->use_graph && b->stream != nullptr && steps > 0;
  if (graph) {
    rc = ensure_graph(b, in, out, frames_in_ring, steps, false);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  const auto h0 = std::chrono::steady_clock::now();
  if (graph) {
    rc = launch_graphs(b);
    if (rc) return rc;
  } else {
    rc = fused_steps(b, in, out, frames_in_ring, steps);
    if (rc) return rc;
  }
  b->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  HIP_TRY(hipEventSynchronize(b->ev1));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
  return flow_check(b);
}

int AspNsBatch_ExportState(AspNsBatch* b, int stream, AspNsState* out) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!out || stream < 0 || stream >= b->S) return fail(ASP_ERR_PARAM, "ExportState: bad argument");
  std::vector<float> blk(kStreamDwords);
  std::vector<int32_t> hh(kHistDwords);
  HIP_TRY(hipStreamSynchronize(b->stream));
  rc = flow_check(b);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(blk.data(), b->state + (size_t)stream * kStreamDwords, kStreamDwords * 4,
                    hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hh.data(), b->hist + (size_t)stream * kHistDwords, kHistDwords * 4,
                    hipMemcpyDeviceToHost));
  unpack_stream(blk.data(), hh.data(), b->paired, out);
  return ASP_OK;
}

int AspNsBatch_ImportState(AspNsBatch* b, int stream, const AspNsState* in) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in || stream < 0 || stream >= b->S) return fail(ASP_ERR_PARAM, "ImportState: bad argument");
  if ((uint32_t)in->fs != b->fs)
    return fail(ASP_ERR_PARAM, "ImportState: the state's fs differs from the batch's (AspNsBatch_Init)");
  for (int i = geo_anal(in->fs) - geo_block(in->fs); i < geo_anal(in->fs); ++i)
    if (in->syntBuf[i] != 0.f)
      return fail(ASP_ERR_PARAM, "ImportState: syntBuf beyond the carried overlap must be zero (between frames)");
  if (!state_is_paired(in)) {
    rc = ensure_unpaired(b);
    if (rc) return rc;
  }
  std::vector<float> blk(kStreamDwords);
  std::vector<int32_t> hh(kHistDwords);
  pack_stream(in, blk.data(), hh.data());
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(b->state + (size_t)stream * kStreamDwords, blk.data(), kStreamDwords * 4,
                    hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->hist + (size_t)stream * kHistDwords, hh.data(), kHistDwords * 4,
                    hipMemcpyHostToDevice));
  return ASP_OK;
}

int AspNsBatch_prior_speech_probability(AspNsBatch* b, float* out) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!out) return fail(ASP_ERR_PARAM, "null output");
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy2D(out, sizeof(float), b->state + kOffScalars + S_PRIORSPEECHPROB,
                      (size_t)kStreamDwords * 4, sizeof(float), (size_t)b->S,
                      hipMemcpyDeviceToHost));
  return ASP_OK;
}

int AspNsBatch_SetSplit(AspNsBatch* b, int parts) {
  if (!b || parts < 1 || parts > 4) return fail(ASP_ERR_PARAM, "SetSplit: parts must be 1..4");
  HIP_TRY(hipSetDevice(b->device));
  if (!b->fork_ev) HIP_TRY(hipEventCreateWithFlags(&b->fork_ev, hipEventDisableTiming));
  for (int i = 0; i < parts - 1; ++i) {
    if (!b->side[i]) HIP_TRY(hipStreamCreateWithFlags(&b->side[i], hipStreamNonBlocking));
    if (!b->join_ev[i]) HIP_TRY(hipEventCreateWithFlags(&b->join_ev[i], hipEventDisableTiming));
  }
  b->split = parts;
  return ASP_OK;
}

// Diagnostic: one fused step of the pair-layout kernel with the phase stamps of workgroup 0's first wave (16 values).
int AspNsBatch_DebugStamps(AspNsBatch* b, const float* in_dev, float* out_dev,
                           unsigned long long* stamps16) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in_dev || !out_dev || !stamps16 || !b->paired || b->kernel == 1 || b->fs == 8000)
    return fail(ASP_ERR_PARAM, "DebugStamps: bad argument (paired state, pair-layout kernel)");
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 16 * sizeof(unsigned long long)));
  hipError_t e = hipMemset(d, 0, 16 * sizeof(unsigned long long));
  if (e == hipSuccess)
    e = launch_ns_frame1(false, b->state, b->hist, b->tables, in_dev, out_dev, b->S, b->stream, d);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e == hipSuccess) e = hipMemcpy(stamps16, d, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "DebugStamps", e);
  return ASP_OK;
}

// Diagnostic: `steps` frame steps of the hand-off build in one launch, with the 17 phase stamps (shader clock)
// of one wave in the middle of the launch (x = grid / 2, y = steps / 2: the steady state, neighbours in every
// phase): kernel start, the 15 phase marks of the frame step, stores drained.
int AspNsBatch_DebugFlowStamps(AspNsBatch* b, const float* in_dev, float* out_dev, int frames_in_ring, int steps,
                               unsigned long long* stamps17) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in_dev || !out_dev || !stamps17 || steps < 2 || steps > kFlowMaxSteps || !flow_applies(b, steps))
    return fail(ASP_ERR_PARAM, "DebugFlowStamps: bad argument (hand-off build, 2..64 steps)");
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 17 * sizeof(unsigned long long)));
  hipError_t e = hipMemset(d, 0, 17 * sizeof(unsigned long long));
  b->flow_stamps = d;
  if (e == hipSuccess) rc = flow_steps(b, in_dev, out_dev, frames_in_ring, steps, false);
  b->flow_stamps = nullptr;
  if (e == hipSuccess && rc == ASP_OK) e = hipStreamSynchronize(b->stream);
  if (e == hipSuccess && rc == ASP_OK) e = hipMemcpy(stamps17, d, 17 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (rc) return rc;
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "DebugFlowStamps", e);
  return flow_check(b);
}

// Diagnostic: `steps` fused steps through the product launch path (chains as set by SetSplit) with every
// workgroup of the pair-layout kernel recording four real-time stamps (100 MHz counter: start, first
// loads in, before the last stores, end); out[num_workgroups][4] holds those of the last step.
int AspNsBatch_DebugTimeline(AspNsBatch* b, const float* in_dev, float* out_dev, int frames_in_ring,
                             int steps, unsigned long long* out, int num_workgroups) {
  DeviceScope dev_scope_;
  int rc = check(b);
  if (rc) return rc;
  if (!in_dev || !out_dev || !out || steps <= 0 || frames_in_ring <= 0 || !b->paired || b->kernel == 1 ||
      b->fs == 8000 || num_workgroups != (b->S + 3) / 4 || (b->S & 3))
    return fail(ASP_ERR_PARAM, "DebugTimeline: bad argument (pair-layout kernel, stream count a multiple of 4)");
  const size_t bytes = (size_t)num_workgroups * 4 * sizeof(unsigned long long);
  HIP_TRY(hipMalloc((void**)&b->timeline, bytes));
  hipError_t e = hipMemset(b->timeline, 0, bytes);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e == hipSuccess) {
    rc = fused_steps(b, in_dev, out_dev, frames_in_ring, steps);
    if (rc == ASP_OK) e = hipStreamSynchronize(b->stream);
  }
  if (e == hipSuccess && rc == ASP_OK) e = hipMemcpy(out, b->timeline, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(b->timeline);
  b->timeline = nullptr;
  if (rc) return rc;
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "DebugTimeline", e);
  return ASP_OK;
}

// Host time (us) the last AspNsBatch_TimedSteps call spent enqueuing its launches (all steps): the
// per-rank host budget of a multi-GPU node, printed by bench.py beside the device time.
int AspNsBatch_LastEnqueueUs(AspNsBatch* b, double* us) {
  if (!b || !us) return fail(ASP_ERR_PARAM, "LastEnqueueUs: bad argument");
  *us = b->last_enqueue_us;
  return ASP_OK;
}

int AspNsBatch_SetFlow(AspNsBatch* b, int mode) {
  if (!b || mode < -1 || mode > 1) return fail(ASP_ERR_PARAM, "SetFlow: -1 (default), 0 (off) or 1 (on)");
  b->flow = mode;
  return ASP_OK;
}

// Diagnostic (tests only): make the next hand-off launches wait for a step that never ran, so that the
// bounded wait, the abort word and the error path can be exercised: the host's step counter moves one
// ahead of the device's.
int AspNsBatch_DebugFlowDesync(AspNsBatch* b) {
  if (!b) return fail(ASP_ERR_PARAM, "null batch handle");
  b->flow_count++;
  return ASP_OK;
}

int AspNsBatch_SetGraph(AspNsBatch* b, int on) {
  if (!b) return fail(ASP_ERR_PARAM, "null batch handle");
  b->use_graph = (on & 1) != 0;
  return ASP_OK;
}

int AspNsBatch_SetKernel(AspNsBatch* b, int kernel) {
  if (!b || kernel < 0 || kernel > 3)
    return fail(ASP_ERR_PARAM, "SetKernel: 0 (default), 1 (one stream per wave, bins q / q + 64), 2 (two streams per wave) or 3 (one stream per wave, pair layout)");
  b->kernel = kernel;
  return ASP_OK;
}

int AspNsBatch_SetStream(AspNsBatch* b, void* hip_stream) {
  if (!b) return fail(ASP_ERR_PARAM, "null batch handle");
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
  b->stream = (hipStream_t)hip_stream;
  b->own_stream = false;
  return ASP_OK;
}

void* AspNsBatch_GetStream(AspNsBatch* b) { return b ? (void*)b->stream : nullptr; }

int AspNsBatch_Synchronize(AspNsBatch* b) {
  if (!b) return fail(ASP_ERR_PARAM, "null batch handle");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return flow_check(b);
}

// ---- streaming-copy ceiling of the box (bench.py prints it next to the 8 TB/s spec peak) ----
__global__ __launch_bounds__(256) void asp_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  // 4 independent 16-byte loads per thread in flight, then 4 stores; consecutive threads touch
  // consecutive float4s of each of the four slices a block owns
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t i = base + (size_t)k * 256;
    v[k] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i < n4) dst[i] = v[k];
  }
}

// Copies `bytes` (a multiple of 16) src -> dst `iters` times on `device`; *gbps = (read + written
// bytes) / hipEvent time of the best of three timed passes.
int AspNs_CopyCeiling(size_t bytes, int iters, int device, double* gbps) {
  if (!gbps || bytes < 4096 || (bytes & 15) || iters < 1) return fail(ASP_ERR_PARAM, "CopyCeiling: bad argument");
  int rc = select_device(device);
  if (rc) return rc;
  float4 *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc((void**)&src, bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&dst, bytes);
  if (e == hipSuccess) e = hipMemset(src, 0x3c, bytes);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  const size_t n4 = bytes / 16;
  const dim3 grid((unsigned)((n4 + 1023) / 1024)), block(256);
  double best = 0.0;
  for (int pass = 0; pass < 4 && e == hipSuccess; ++pass) {
    e = hipEventRecord(e0, nullptr);
    for (int k = 0; k < iters; ++k) hipLaunchKernelGGL(asp_copy_kernel, grid, block, 0, nullptr, src, dst, n4);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && pass > 0 && ms > 0.f) {
      const double g = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
      if (g > best) best = g;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "CopyCeiling", e);
  *gbps = best;
  return ASP_OK;
}

int AspNs_DeviceAlloc(void** ptr, size_t bytes, int device) {
  if (!ptr) return fail(ASP_ERR_PARAM, "null pointer");
  int rc = select_device(device);
  if (rc) return rc;
  HIP_TRY(hipMalloc(ptr, bytes));
  return ASP_OK;
}
int AspNs_DeviceFree(void* ptr) {
  HIP_TRY(hipFree(ptr));
  return ASP_OK;
}
int AspNs_MemcpyH2D(void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return ASP_OK;
}
int AspNs_MemcpyD2H(void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return ASP_OK;
}

static int rdft_batch(float* data, int count, int isgn, int mem, int device, int n) {
  if (!data || count <= 0) return fail(ASP_ERR_PARAM, "rdft batch: bad argument");
  int rc = select_device(device);
  if (rc) return rc;
  NsTables* T = nullptr;
  rc = device_tables(device, &T);
  if (rc) return rc;
  float* d = data;
  const size_t bytes = (size_t)count * n * sizeof(float);
  if (mem == ASP_MEM_HOST) {
    HIP_TRY(hipMalloc((void**)&d, bytes));
    hipError_t e0 = hipMemcpy(d, data, bytes, hipMemcpyHostToDevice);
    if (e0 != hipSuccess) {
      (void)hipFree(d);
      return fail(ASP_ERR_HIP, "rdft batch: upload", e0);
    }
  }
  hipError_t e = launch_rdft256(d, count, isgn, T, nullptr, n);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (mem == ASP_MEM_HOST) {
    if (e == hipSuccess) e = hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
  }
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "rdft batch", e);
  return ASP_OK;
}
int AspNs_rdft256_batch(float* data, int count, int isgn, int mem, int device) {
  return rdft_batch(data, count, isgn, mem, device, 256);
}
int AspNs_rdft128_batch(float* data, int count, int isgn, int mem, int device) {
  return rdft_batch(data, count, isgn, mem, device, 128);
}

// Test seams for the device math (see debug_fn in ns_kernels.hip).
int AspNs_debug_eval(int fn, float* data, size_t n, int device) {
  if (!data || n == 0) return fail(ASP_ERR_PARAM, "debug_eval: bad argument");
  int rc = select_device(device);
  if (rc) return rc;
  NsTables* T = nullptr;
  rc = device_tables(device, &T);
  if (rc) return rc;
  float* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, n * sizeof(float)));
  hipError_t e = hipMemcpy(d, data, n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_debug_eval(fn, d, n, T, nullptr);
  if (e == hipSuccess) e = hipMemcpy(data, d, n * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "debug_eval", e);
  return ASP_OK;
}

int AspNs_debug_compare(int fn_a, int fn_b, uint32_t start, uint32_t count, uint32_t* n_bad,
                        uint32_t* bad_bits64, float param, int device) {
  if (!n_bad || !bad_bits64) return fail(ASP_ERR_PARAM, "debug_compare: bad argument");
  int rc = select_device(device);
  if (rc) return rc;
  NsTables* T = nullptr;
  rc = device_tables(device, &T);
  if (rc) return rc;
  unsigned* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 65 * sizeof(unsigned)));
  hipError_t e = hipMemset(d, 0, 65 * sizeof(unsigned));
  if (e == hipSuccess) e = launch_debug_compare(fn_a, fn_b, start, count, d, d + 1, param, T, nullptr);
  unsigned h[65];
  if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(ASP_ERR_HIP, "debug_compare", e);
  *n_bad = h[0];
  memcpy(bad_bits64, h + 1, 64 * sizeof(unsigned));
  return ASP_OK;
}

// ----------------------------------------------------------------- layer 1
// The reference's per-stream API (ns/noise_suppression.c:20-66) over a batch of
// one stream on device 0.  Frames are host pointers, as in the reference.

struct NsHandleT {
  AspNsBatch* batch;
  int initFlag;
};

int WebRtcNs_Create(NsHandle** NS_inst) {
  if (!NS_inst) return -1;
  NsHandleT* h = (NsHandleT*)malloc(sizeof(NsHandleT));
  if (!h) return -1;
  h->batch = nullptr;
  h->initFlag = 0;
  if (AspNsBatch_Create(&h->batch, 1, 0) != ASP_OK) {
    fprintf(stderr, "WebRtcNs_Create: %s\n", g_err);
    free(h);
    *NS_inst = NULL;
    return -1;
  }
  *NS_inst = h;
  return 0;
}

int WebRtcNs_Free(NsHandle* NS_inst) {
  if (NS_inst) {
    AspNsBatch_Free(NS_inst->batch);
    free(NS_inst);
  }
  return 0;
}

int WebRtcNs_Init(NsHandle* NS_inst, uint32_t fs) {
  if (!NS_inst) return -1;
  if (AspNsBatch_Init(NS_inst->batch, fs) != ASP_OK) return -1;
  NS_inst->initFlag = 1;
  return 0;
}

int WebRtcNs_set_policy(NsHandle* NS_inst, int mode) {
  if (!NS_inst || !NS_inst->initFlag) return -1;
  return AspNsBatch_set_policy(NS_inst->batch, mode) == ASP_OK ? 0 : -1;
}

void WebRtcNs_Analyze(NsHandle* NS_inst, const float* spframe) {
  if (!NS_inst || !NS_inst->initFlag) {  // reference: assert(initFlag == 1), ns_core.c:1064
    fprintf(stderr, "WebRtcNs_Analyze: handle not initialised\n");
    abort();
  }
  if (AspNsBatch_Analyze(NS_inst->batch, spframe, ASP_MEM_HOST) != ASP_OK) {
    fprintf(stderr, "WebRtcNs_Analyze: %s\n", g_err);
    abort();
  }
}

void WebRtcNs_Process(NsHandle* NS_inst, const float* const* spframe, int num_bands,
                      float* const* outframe) {
  if (!NS_inst || !NS_inst->initFlag) {  // ns_core.c:1208
    fprintf(stderr, "WebRtcNs_Process: handle not initialised\n");
    abort();
  }
  if (num_bands != AspNsBatch_num_bands(NS_inst->batch)) {  // the reference trusts the caller here
    fprintf(stderr, "WebRtcNs_Process: num_bands = %d does not match the rate given to WebRtcNs_Init\n",
            num_bands);
    abort();
  }
  if (num_bands > 1) {
    float hin[2 * kBlockL], hout[2 * kBlockL];
    for (int k = 1; k < num_bands; ++k) memcpy(hin + (k - 1) * kBlockL, spframe[k], kBlockL * sizeof(float));
    if (AspNsBatch_ProcessBands(NS_inst->batch, spframe[0], hin, outframe[0], hout, ASP_MEM_HOST) != ASP_OK) {
      fprintf(stderr, "WebRtcNs_Process: %s\n", g_err);
      abort();
    }
    for (int k = 1; k < num_bands; ++k) memcpy(outframe[k], hout + (k - 1) * kBlockL, kBlockL * sizeof(float));
    return;
  }
  if (AspNsBatch_Process(NS_inst->batch, spframe[0], outframe[0], ASP_MEM_HOST) != ASP_OK) {
    fprintf(stderr, "WebRtcNs_Process: %s\n", g_err);
    abort();
  }
}

float WebRtcNs_prior_speech_probability(NsHandle* handle) {
  if (handle == NULL) return -1;
  if (handle->initFlag == 0) return -1;
  float p = -1.f;
  if (AspNsBatch_prior_speech_probability(handle->batch, &p) != ASP_OK) return -1;
  return p;
}

}  // extern "C"
