// bt_api.hip -- host side of include/asp_bt.h: constant tables, the batch handle, and the
// reference's per-stream blockThreshold_* protocol (audioDenoiseBlockTreshold.h:46-74)
// implemented as a batch of one with host-side hop buffering.  No CPU fallback.
#include <hip/hip_runtime.h>

#include "device_scope.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "asp_bt.h"
#include "asp_ns.h"
#include "bt_layout.h"

using namespace aspbt;

namespace aspbt {
size_t macroblock_lds_bytes(int n);
hipError_t launch_bt_macroblock(int n, float* state, const BtTables* T, const float* in,
                                float* out, int num_streams, int frames, int threshold,
                                int in_stride, int out_stride, hipStream_t s,
                                unsigned long long* stamps = nullptr);
hipError_t launch_bt_fftr(int n, const float* src, float* dst, int count, int inverse,
                          const BtTables* T, hipStream_t s);
hipError_t launch_bt_macroblock8_flow(bool q4, float* state, const BtTables* T, const float* in, float* out, int num_streams,
                                      int stride, hipStream_t s, unsigned* seq, unsigned* abort_w, unsigned want, int steps,
                                      int slot0, int ring, size_t per);
hipError_t launch_bt_macroblock_any(const BtAnyTables& A, float* state, const float* in, float* out, int num_streams,
                                    int frames, int threshold, int in_stride, int out_stride, hipStream_t s,
                                    unsigned long long* stamps = nullptr);
hipError_t launch_bt_fftr_any(const BtAnyTables& A, const float* src, float* dst, int count, int inverse, hipStream_t s);
}  // namespace aspbt

namespace {

thread_local char g_bt_err[512] = "";
int bt_fail(int code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess)
    snprintf(g_bt_err, sizeof g_bt_err, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(g_bt_err, sizeof g_bt_err, "%s", what);
  fprintf(stderr, "asp_bt: %s\n", g_bt_err);
  return code;
}
#define BT_TRY(expr)                                            \
  do {                                                          \
    hipError_t e_ = (expr);                                     \
    if (e_ != hipSuccess) return bt_fail(ASP_ERR_HIP, #expr, e_); \
  } while (0)

// NOTE: C++ translation unit; every libm call casts to double explicitly so the
// arithmetic is that of the reference's C (see ns_api.hip).
void fill_size(BtSize* S, int n, float* hann, float* tw_f, float* tw_i, float* sup_f, float* sup_i) {
  const int half = n / 2, nc = n / 2;
  for (int i = 0; i < half; i++) {  // make_hanning_window, .c:70-77
    hann[i] = (float)(0.5 - 0.5 * cos(2 * M_PI * i / (double)(n - 1)));
    hann[n - 1 - i] = hann[i];
  }
  for (int i = 0; i < nc; ++i) {  // kiss_fft_alloc, kiss_fft.c:357-363
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    double phase = -2 * pi * i / nc;
    tw_f[2 * i] = (float)cos(phase);
    tw_f[2 * i + 1] = (float)sin(phase);
    phase *= -1;
    tw_i[2 * i] = (float)cos(phase);
    tw_i[2 * i + 1] = (float)sin(phase);
  }
  for (int i = 0; i < nc / 2; ++i) {  // kiss_fftr_alloc, kiss_fftr.c:57-63
    double phase = -3.14159265358979323846264338327 * ((double)(i + 1) / nc + .5);
    sup_f[2 * i] = (float)cos(phase);
    sup_f[2 * i + 1] = (float)sin(phase);
    phase *= -1;
    sup_i[2 * i] = (float)cos(phase);
    sup_i[2 * i + 1] = (float)sin(phase);
  }
  const float sigma_noise = (float)0.047;                              // .c:111
  const float sigma_h = (float)((double)sigma_noise * sqrt(0.375));    // .c:112
  static const float m_lambda[3][5] = {{1.5, 1.8, 2, 2.5, 2.5},        // .c:11-13
                                       {1.8, 2, 2.5, 3.5, 3.5},
                                       {2, 2.5, 3.5, 4.7, 4.7}};
  S->norm = (float)(sqrt(2.0) / (sqrt((double)n) * (double)sigma_h));  // .c:365
  const float L_pi = 8.0, Lambda_pi = 2.5;
  S->dc_const = Lambda_pi * L_pi * (sigma_h * sigma_h) * (float)n;     // .c:503
  S->wiener_c = (float)n * (sigma_h * sigma_h);                        // .c:481
  S->pad = 0.f;
  for (int T = 0; T < 3; ++T)
    for (int F = 0; F < 5; ++F) {
      const int TT = 8 >> T, FF = 16 >> F;
      const float lambda = m_lambda[T][F];
      const float size_blk = (float)(TT * FF);
      BtSeg& g = S->seg[T][F];
      g.size_blk = size_blk;
      g.temp = (lambda * lambda) * (size_blk * size_blk) - 2 * lambda * size_blk * (size_blk - 2);
      g.thr = lambda * size_blk;
      g.two_size = 2 * size_blk;
      g.a_const = (float)((double)(lambda * TT * FF) * pow((double)sigma_h, 2.0) * (double)n);
    }
}

// false: the inverse twiddles are not the exact conjugates bt_kernels8.hip assumes (an internal error)
bool build_bt_tables(BtTables* T) {
  memset(T, 0, sizeof *T);
  fill_size(&T->s256, 256, T->hann256, T->tw256_f, T->tw256_i, T->sup256_f, T->sup256_i);
  fill_size(&T->s1024, 1024, T->hann1024, T->tw1024_f, T->tw1024_i, T->sup1024_f, T->sup1024_i);
  {  // lane terms of the three register exchanges of bt_kernels8.hip
    const Lay* lay[4] = {&LA, &LB, &LC, &LD};
    const Swz* sw[3] = {&S1, &S2, &S3};
    for (int x = 0; x < 3; ++x)
      for (int lane = 0; lane < 64; ++lane) {
        T->xterm[2 * x][lane] = (uint16_t)(8 * swz(*sw[x], pos_lane(*lay[x], lane)));
        T->xterm[2 * x + 1][lane] = (uint16_t)(8 * swz(*sw[x], pos_lane(*lay[x + 1], lane)));
      }
  }
  // bt_kernels8.hip derives the inverse twiddles from the forward tables: they must be exact conjugates
  for (int i = 0; i < 512; ++i)
    if (T->tw1024_i[2 * i] != T->tw1024_f[2 * i] || T->tw1024_i[2 * i + 1] != -T->tw1024_f[2 * i + 1]) return false;
  for (int i = 0; i < 256; ++i)
    if (T->sup1024_i[2 * i] != T->sup1024_f[2 * i] || T->sup1024_i[2 * i + 1] != -T->sup1024_f[2 * i + 1]) return false;
  return true;
}

// Tables of a window other than 256 / 1024 samples (bt_macroblock_any_kernel): one device block per batch.
// kf_factor (kiss_fft.c:308-330) and the decimation order of kf_work (kiss_fft.c:237-302) run here.
bool bt_any_window_ok(int win) {
  if (win < 4 || win > kAnyMaxWin || (win & 1)) return false;
  int n = win / 2, p = 4;
  const double floor_sqrt = floor(sqrt((double)n));
  do {  // the generic butterfly's scratch holds kAnyMaxRadix points
    while (n % p) {
      p = p == 4 ? 2 : p == 2 ? 3 : p + 2;
      if (p > floor_sqrt) p = n;
    }
    if (p > kAnyMaxRadix) return false;
    n /= p;
  } while (n > 1);
  return true;
}

int bt_build_any(int win, BtAnyTables* A, void** dev_block) {
  const int nc = win / 2;
  memset(A, 0, sizeof *A);
  A->n = win;
  A->nc = nc;
  A->ncol = (win - 1) / 2 / ASP_BT_NBLK_FREQ;  // .c:493
  {
    int n = nc, p = 4, k = 0;
    const double floor_sqrt = floor(sqrt((double)nc));
    do {
      while (n % p) {
        p = p == 4 ? 2 : p == 2 ? 3 : p + 2;
        if (p > floor_sqrt) p = n;
      }
      n /= p;
      if (k >= 32) return bt_fail(ASP_ERR_PARAM, "window has too many factors");
      A->fac[k++] = p;
      A->fac[k++] = n;
    } while (n > 1);
    A->nfac = k / 2;
  }
  // one host block: hann[win] | tw_f[2 nc] | tw_i[2 nc] | sup_f[2 (nc/2)] | sup_i[2 (nc/2)] | perm (uint16)[nc]
  const size_t nf = (size_t)win + 4 * (size_t)nc + 4 * (size_t)(nc / 2);
  const size_t bytes = nf * sizeof(float) + (size_t)nc * sizeof(uint16_t);
  std::vector<unsigned char> host(bytes);
  float* hf = reinterpret_cast<float*>(host.data());
  float *hann = hf, *tw_f = hann + win, *tw_i = tw_f + 2 * nc, *sup_f = tw_i + 2 * nc, *sup_i = sup_f + 2 * (nc / 2);
  uint16_t* perm = reinterpret_cast<uint16_t*>(sup_i + 2 * (nc / 2));
  fill_size(&A->P, win, hann, tw_f, tw_i, sup_f, sup_i);
  for (int n = 0; n < nc; ++n) {  // input n = sum k_s fstride_s lands at sum k_s m_s
    int t = n, pos = 0;
    for (int q = 0; q < A->nfac; ++q) {
      pos += (t % A->fac[2 * q]) * A->fac[2 * q + 1];
      t /= A->fac[2 * q];
    }
    perm[n] = (uint16_t)pos;
  }
  // device block: the struct itself (256 B aligned head), then the tables
  const size_t head = (sizeof(BtAnyTables) + 255) / 256 * 256;
  unsigned char* dev = nullptr;
  BT_TRY(hipMalloc((void**)&dev, head + bytes));
  hipError_t e = hipMemcpy(dev + head, host.data(), bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(dev);
    return bt_fail(ASP_ERR_HIP, "uploading the window's tables", e);
  }
  float* df = reinterpret_cast<float*>(dev + head);
  A->hann = df;
  A->tw_f = df + win;
  A->tw_i = A->tw_f + 2 * nc;
  A->sup_f = A->tw_i + 2 * nc;
  A->sup_i = A->sup_f + 2 * (nc / 2);
  A->perm = reinterpret_cast<const uint16_t*>(A->sup_i + 2 * (nc / 2));
  A->self = reinterpret_cast<const BtAnyTables*>(dev);
  e = hipMemcpy(dev, A, sizeof(BtAnyTables), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(dev);
    return bt_fail(ASP_ERR_HIP, "uploading the window's plan", e);
  }
  *dev_block = dev;
  return ASP_OK;
}

std::mutex g_bt_mu;
BtTables* g_bt_dev[64] = {nullptr};

int bt_tables(int device, BtTables** out) {
  if (device < 0 || device >= 64) return bt_fail(ASP_ERR_PARAM, "device ordinal out of range");
  std::lock_guard<std::mutex> lk(g_bt_mu);
  if (!g_bt_dev[device]) {
    BtTables* host = (BtTables*)malloc(sizeof(BtTables));
    if (!host) return bt_fail(ASP_ERR_STATE, "BT tables: out of host memory");
    if (!build_bt_tables(host)) {
      free(host);
      return bt_fail(ASP_ERR_STATE, "BT tables: inverse twiddles are not the conjugates of the forward ones");
    }
    BtTables* dev = nullptr;
    hipError_t e = hipMalloc((void**)&dev, sizeof(BtTables));
    if (e == hipSuccess) e = hipMemcpy(dev, host, sizeof(BtTables), hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess) return bt_fail(ASP_ERR_HIP, "uploading BT tables", e);
    g_bt_dev[device] = dev;
  }
  *out = g_bt_dev[device];
  return ASP_OK;
}

int bt_select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return bt_fail(ASP_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)", e);
  if (device < 0 || device >= n) return bt_fail(ASP_ERR_PARAM, "device ordinal out of range");
  BT_TRY(hipSetDevice(device));
  return ASP_OK;
}

}  // namespace

struct AspBtBatch {
  int S = 0, win = 0, half = 0, macro = 0, device = 0;
  hipStream_t stream = nullptr;
  float* state = nullptr;
  int state_floats = kStateFloats, off_out = kOffOutTail;  // per-stream state block (kAny* for other windows)
  BtTables* tables = nullptr;
  bool any = false;            // a window other than 256 / 1024 samples: bt_macroblock_any_kernel
  BtAnyTables any_tables = {};
  void* any_block = nullptr;   // the device block any_tables points into
  float *stage_in = nullptr, *stage_out = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // second launch chain of the K-step path (AspBtBatch_TimedSteps): stream-channels are independent, so
  // the two halves of a large batch run as two chains whose launch boundaries overlap
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // Hand-off build of the multi-macroblock entry points (AspBtBatch_DenoiseBlocks, TimedSteps; bt_kernels8.hip,
  // BtFlowArgs): up to 64 consecutive macroblocks of every stream-channel per launch, a per-stream counter in memory
  // orders a stream-channel's overlap-add tails.  -1 = default (on for 256 / 1024-sample windows), 0 = off, 1 = on.
  int flow = -1;
  unsigned* flow_seq = nullptr;    // [S]
  unsigned* flow_abort = nullptr;  // 16 B
  unsigned flow_count = 0;
  bool flow_unchecked = false;
};

namespace {
bool bt_flow_applies(const AspBtBatch* b, int steps, const float* in, float* out) {
  const char* e = getenv("ASP_BT_FLOW");
  const bool on = b->flow < 0 ? !(e && e[0] == '0') : b->flow != 0;
  // the tuned kernels' windows, 8-byte aligned frames (their 8-byte accesses), whole groups of four at 256 samples
  return on && steps >= 2 && !b->any && (b->win == 1024 || (b->win == 256 && b->S % 4 == 0)) &&
         ((uintptr_t)in & 7) == 0 && ((uintptr_t)out & 7) == 0;
}
int bt_flow_steps(AspBtBatch* b, const float* in, float* out, int ring, int steps) {
  if (!b->flow_seq) {
    BT_TRY(hipMalloc((void**)&b->flow_seq, (size_t)b->S * sizeof(unsigned)));
    BT_TRY(hipMalloc((void**)&b->flow_abort, 16));
    BT_TRY(hipMemsetAsync(b->flow_seq, 0, (size_t)b->S * sizeof(unsigned), b->stream));
    BT_TRY(hipMemsetAsync(b->flow_abort, 0, 16, b->stream));
    b->flow_count = 0;
  }
  const size_t per = (size_t)b->S * b->macro;
  for (int k = 0; k < steps; k += 64) {
    const int m = steps - k < 64 ? steps - k : 64;
    BT_TRY(launch_bt_macroblock8_flow(b->win == 256, b->state, b->tables, in, out, b->S, b->macro, b->stream, b->flow_seq,
                                      b->flow_abort, b->flow_count, m, k % ring, ring, per));
    b->flow_count += (unsigned)m;
  }
  b->flow_unchecked = true;
  return ASP_OK;
}
// after the batch's stream has been synchronised: did a hand-off wait time out?
int bt_flow_check(AspBtBatch* b) {
  if (!b->flow_unchecked) return ASP_OK;
  b->flow_unchecked = false;
  unsigned a = 0;
  BT_TRY(hipMemcpy(&a, b->flow_abort, sizeof a, hipMemcpyDeviceToHost));
  if (a == 0) return ASP_OK;
  std::vector<unsigned> seq((size_t)b->S, b->flow_count);
  BT_TRY(hipMemcpy(b->flow_seq, seq.data(), seq.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  BT_TRY(hipMemset(b->flow_abort, 0, 16));
  return bt_fail(ASP_ERR_HIP, "BT hand-off wait timed out: overlap-add tails were skipped, reset the batch");
}
}  // namespace

extern "C" {

int AspBtBatch_Create(AspBtBatch** out, int num_streams, int win_size, int device) {
  AspDeviceScope dev_scope_;
  if (!out || num_streams <= 0) return bt_fail(ASP_ERR_PARAM, "AspBtBatch_Create: bad argument");
  *out = nullptr;
  const bool any = win_size != 256 && win_size != 1024;
  if (any && !bt_any_window_ok(win_size))
    return bt_fail(ASP_ERR_PARAM, "AspBtBatch_Create: win_size must be even, 4 .. 2048, with no prime factor of win_size / 2 above 32");
  int rc = bt_select_device(device);
  if (rc) return rc;
  AspBtBatch* b = new AspBtBatch();
  b->S = num_streams;
  b->win = win_size;
  b->half = win_size / 2;
  b->macro = 8 * b->half;
  b->device = device;
  b->any = any;
  if (any) b->state_floats = kAnyStateFloats, b->off_out = kAnyOffOutTail;
  rc = any ? bt_build_any(win_size, &b->any_tables, &b->any_block) : bt_tables(device, &b->tables);
  if (rc) {
    delete b;
    return rc;
  }
  const size_t frame_bytes = (size_t)num_streams * b->macro * sizeof(float);
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&b->state, (size_t)num_streams * b->state_floats * 4);
  if (e == hipSuccess) e = hipMemsetAsync(b->state, 0, (size_t)num_streams * b->state_floats * 4, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_in, frame_bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_out, frame_bytes);
  if (e == hipSuccess) e = hipEventCreate(&b->ev0);
  if (e == hipSuccess) e = hipEventCreate(&b->ev1);
  if (e != hipSuccess) {
    AspBtBatch_Free(b);
    return bt_fail(ASP_ERR_HIP, "AspBtBatch_Create: device allocation", e);
  }
  *out = b;
  return ASP_OK;
}

int AspBtBatch_Free(AspBtBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return ASP_OK;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->state) (void)hipFree(b->state);
  if (b->any_block) (void)hipFree(b->any_block);
  if (b->flow_seq) (void)hipFree(b->flow_seq);
  if (b->flow_abort) (void)hipFree(b->flow_abort);
  if (b->stage_in) (void)hipFree(b->stage_in);
  if (b->stage_out) (void)hipFree(b->stage_out);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->side) {
    (void)hipStreamSynchronize(b->side);
    (void)hipStreamDestroy(b->side);
    (void)hipEventDestroy(b->ev_fork);
    (void)hipEventDestroy(b->ev_join);
  }
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return ASP_OK;
}

int AspBtBatch_Reset(AspBtBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return bt_fail(ASP_ERR_PARAM, "null batch handle");
  BT_TRY(hipSetDevice(b->device));
  BT_TRY(hipMemsetAsync(b->state, 0, (size_t)b->S * b->state_floats * 4, b->stream));
  return ASP_OK;
}

// blockThreshold_reset of ONE stream-channel of a running batch (audioDenoiseBlockTreshold.c:692-707 per handle):
// its input tail and overlap-add tail are cleared, the others untouched.
int AspBtBatch_ResetStream(AspBtBatch* b, int stream) {
  AspDeviceScope dev_scope_;
  if (!b) return bt_fail(ASP_ERR_PARAM, "null batch handle");
  if (stream < 0 || stream >= b->S) return bt_fail(ASP_ERR_PARAM, "ResetStream: stream out of range");
  BT_TRY(hipSetDevice(b->device));
  BT_TRY(hipMemsetAsync(b->state + (size_t)stream * b->state_floats, 0, (size_t)b->state_floats * 4, b->stream));
  return ASP_OK;
}

int AspBtBatch_num_streams(const AspBtBatch* b) { return b ? b->S : ASP_ERR_PARAM; }
int AspBtBatch_macro_size(const AspBtBatch* b) { return b ? b->macro : ASP_ERR_PARAM; }

static hipError_t bt_launch(AspBtBatch* b, float* state, const float* in, float* out, int num_streams, int frames,
                            int threshold, int in_stride, int out_stride, hipStream_t s,
                            unsigned long long* stamps = nullptr) {
  if (b->any)
    return launch_bt_macroblock_any(b->any_tables, state, in, out, num_streams, frames, threshold, in_stride,
                                    out_stride, s, stamps);
  return launch_bt_macroblock(b->win, state, b->tables, in, out, num_streams, frames, threshold, in_stride, out_stride,
                              s, stamps);
}

static int bt_run(AspBtBatch* b, const float* in, float* out, int frames, int threshold, int mem) {
  if (!b || !out || (!in && frames > 0)) return bt_fail(ASP_ERR_PARAM, "null argument");
  if (frames < 0 || frames > 8) return bt_fail(ASP_ERR_PARAM, "frames must be 0..8");
  BT_TRY(hipSetDevice(b->device));
  const int n = frames * b->half;
  const float* din = in;
  float* dout = out;
  if (frames == 0) {
    // a flush with no pending hop emits nothing but still consumes the overlap tail, as the reference does
    // (.c:656-659: outbuf is shifted and cleared whatever have_nblk_time is)
    if (!threshold)
      BT_TRY(hipMemset2DAsync(b->state + b->off_out, (size_t)b->state_floats * 4, 0, (size_t)b->half * 4, (size_t)b->S, b->stream));
    return ASP_OK;
  }
  if (mem == ASP_MEM_HOST) {
    BT_TRY(hipMemcpyAsync(b->stage_in, in, (size_t)b->S * n * 4, hipMemcpyHostToDevice, b->stream));
    din = b->stage_in;
    dout = b->stage_out;
  } else if (mem != ASP_MEM_DEVICE) {
    return bt_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  BT_TRY(bt_launch(b, b->state, din, dout, b->S, frames, threshold, n, n, b->stream));
  if (mem == ASP_MEM_HOST) {
    BT_TRY(hipMemcpyAsync(out, b->stage_out, (size_t)b->S * n * 4, hipMemcpyDeviceToHost,
                          b->stream));
    BT_TRY(hipStreamSynchronize(b->stream));
  }
  return ASP_OK;
}

int AspBtBatch_Denoise(AspBtBatch* b, const float* in, float* out, int mem) {
  AspDeviceScope dev_scope_;
  return bt_run(b, in, out, 8, 1, mem);
}

int AspBtBatch_Flush(AspBtBatch* b, const float* in, int hops, float* out, int mem) {
  AspDeviceScope dev_scope_;
  if (hops < 0 || hops > 7) return bt_fail(ASP_ERR_PARAM, "Flush: hops must be 0..7");
  return bt_run(b, in, out, hops, 0, mem);
}

int AspBtBatch_Synchronize(AspBtBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return bt_fail(ASP_ERR_PARAM, "null batch handle");
  BT_TRY(hipSetDevice(b->device));
  BT_TRY(hipStreamSynchronize(b->stream));
  return bt_flow_check(b);
}

int AspBtBatch_SetFlow(AspBtBatch* b, int mode) {
  if (!b || mode < -1 || mode > 1) return bt_fail(ASP_ERR_PARAM, "SetFlow: -1 (default), 0 (off) or 1 (on)");
  b->flow = mode;
  return ASP_OK;
}

// `nblocks` consecutive macroblocks of every stream-channel: in / out [nblocks][num_streams][macro] (the layout of
// TimedSteps' ring).  The 256 / 1024-sample windows take the hand-off build (up to 64 macroblocks per launch), other
// windows one launch per macroblock.  mem == ASP_MEM_HOST copies in and out around the launches and synchronises.
int AspBtBatch_DenoiseBlocks(AspBtBatch* b, const float* in, float* out, int nblocks, int mem) {
  AspDeviceScope dev_scope_;
  if (!b || !in || !out || nblocks < 0) return bt_fail(ASP_ERR_PARAM, "DenoiseBlocks: bad argument");
  if (mem != ASP_MEM_HOST && mem != ASP_MEM_DEVICE) return bt_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  if (nblocks == 0) return ASP_OK;
  BT_TRY(hipSetDevice(b->device));
  const size_t per = (size_t)b->S * b->macro;
  const float* din = in;
  float* dout = out;
  float *tmp_in = nullptr, *tmp_out = nullptr;
  if (mem == ASP_MEM_HOST) {
    BT_TRY(hipMalloc((void**)&tmp_in, per * nblocks * sizeof(float)));
    hipError_t e = hipMalloc((void**)&tmp_out, per * nblocks * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(tmp_in, in, per * nblocks * sizeof(float), hipMemcpyHostToDevice, b->stream);
    if (e != hipSuccess) {
      (void)hipFree(tmp_in);
      if (tmp_out) (void)hipFree(tmp_out);
      return bt_fail(ASP_ERR_HIP, "DenoiseBlocks: staging", e);
    }
    din = tmp_in;
    dout = tmp_out;
  }
  int rc = ASP_OK;
  if (bt_flow_applies(b, nblocks, din, dout)) {
    rc = bt_flow_steps(b, din, dout, nblocks, nblocks);
  } else {
    for (int k = 0; k < nblocks && rc == ASP_OK; ++k)
      if (bt_launch(b, b->state, din + per * k, dout + per * k, b->S, 8, 1, b->macro, b->macro, b->stream) != hipSuccess)
        rc = bt_fail(ASP_ERR_HIP, "DenoiseBlocks: launch");
  }
  if (mem == ASP_MEM_HOST) {
    hipError_t e = hipSuccess;
    if (rc == ASP_OK) e = hipMemcpyAsync(out, tmp_out, per * nblocks * sizeof(float), hipMemcpyDeviceToHost, b->stream);
    const hipError_t e2 = hipStreamSynchronize(b->stream);
    (void)hipFree(tmp_in);
    (void)hipFree(tmp_out);
    if (rc == ASP_OK && (e != hipSuccess || e2 != hipSuccess)) return bt_fail(ASP_ERR_HIP, "DenoiseBlocks: copy back", e != hipSuccess ? e : e2);
    if (rc == ASP_OK) rc = bt_flow_check(b);
  }
  return rc;
}

int AspBtBatch_TimedSteps(AspBtBatch* b, const float* in, float* out, int blocks_in_ring,
                          int steps, float* elapsed_ms) {
  AspDeviceScope dev_scope_;
  if (!b || !in || !out || blocks_in_ring <= 0 || steps < 0 || !elapsed_ms)
    return bt_fail(ASP_ERR_PARAM, "TimedSteps: bad argument");
  BT_TRY(hipSetDevice(b->device));
  const size_t per = (size_t)b->S * b->macro;
  if (bt_flow_applies(b, steps, in, out)) {
    BT_TRY(hipEventRecord(b->ev0, b->stream));
    const int rcf = bt_flow_steps(b, in, out, blocks_in_ring, steps);
    if (rcf != ASP_OK) return rcf;
    BT_TRY(hipEventRecord(b->ev1, b->stream));
    BT_TRY(hipEventSynchronize(b->ev1));
    BT_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
    return bt_flow_check(b);
  }
  const char* ch = getenv("ASP_BT_CHAINS");
  const bool dual = b->S >= 2048 && !(ch && atoi(ch) == 1);
  if (dual && !b->side) {
    BT_TRY(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
    BT_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
    BT_TRY(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
  }
  BT_TRY(hipEventRecord(b->ev0, b->stream));
  if (dual) {
    BT_TRY(hipEventRecord(b->ev_fork, b->stream));
    BT_TRY(hipStreamWaitEvent(b->side, b->ev_fork, 0));
  }
  const int half = dual ? b->S / 2 : b->S;
  for (int k = 0; k < steps; ++k) {
    const size_t off = per * (size_t)(k % blocks_in_ring);
    BT_TRY(bt_launch(b, b->state, in + off, out + off, half, 8, 1, b->macro, b->macro, b->stream));
    if (dual) {
      const size_t o2 = (size_t)half * b->macro;
      BT_TRY(bt_launch(b, b->state + (size_t)half * b->state_floats, in + off + o2, out + off + o2, b->S - half, 8, 1,
                       b->macro, b->macro, b->side));
    }
  }
  if (dual) {
    BT_TRY(hipEventRecord(b->ev_join, b->side));
    BT_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
  }
  BT_TRY(hipEventRecord(b->ev1, b->stream));
  BT_TRY(hipEventSynchronize(b->ev1));
  BT_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
  return ASP_OK;
}

// Diagnostic: one macroblock launch with phase time stamps of workgroup 0 (s_memtime ticks).
int AspBtBatch_DebugStamps(AspBtBatch* b, const float* in_dev, float* out_dev,
                           unsigned long long* stamps11) {  // 48 slots (0..10 phases, 11..15 sub-phases, 16..47 per-wave stamps of the N = 1024 kernel)
  if (!b || !in_dev || !out_dev || !stamps11) return bt_fail(ASP_ERR_PARAM, "DebugStamps: bad argument");
  BT_TRY(hipSetDevice(b->device));
  unsigned long long* d = nullptr;
  BT_TRY(hipMalloc((void**)&d, 48 * sizeof(unsigned long long)));
  hipError_t e = hipMemset(d, 0, 48 * sizeof(unsigned long long));
  if (e == hipSuccess)
    e = bt_launch(b, b->state, in_dev, out_dev, b->S, 8, 1, b->macro, b->macro, b->stream, d);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e == hipSuccess) e = hipMemcpy(stamps11, d, 48 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return bt_fail(ASP_ERR_HIP, "DebugStamps", e);
  return ASP_OK;
}

int AspBtBatch_ExportState(AspBtBatch* b, int stream, AspBtState* out) {
  AspDeviceScope dev_scope_;
  if (!b || !out || stream < 0 || stream >= b->S) return bt_fail(ASP_ERR_PARAM, "ExportState: bad argument");
  BT_TRY(hipSetDevice(b->device));
  BT_TRY(hipStreamSynchronize(b->stream));
  {
    const int rcf = bt_flow_check(b);
    if (rcf != ASP_OK) return rcf;
  }
  float blk[kAnyStateFloats];
  BT_TRY(hipMemcpy(blk, b->state + (size_t)stream * b->state_floats, sizeof(float) * b->state_floats, hipMemcpyDeviceToHost));
  memset(out, 0, sizeof *out);
  out->win_size = b->win;
  memcpy(out->inbuf_tail, blk + kOffInTail, sizeof(float) * b->half);
  memcpy(out->out_tail, blk + b->off_out, sizeof(float) * b->half);
  return ASP_OK;
}

int AspBtBatch_ImportState(AspBtBatch* b, int stream, const AspBtState* in) {
  AspDeviceScope dev_scope_;
  if (!b || !in || stream < 0 || stream >= b->S) return bt_fail(ASP_ERR_PARAM, "ImportState: bad argument");
  if (in->win_size != b->win) return bt_fail(ASP_ERR_PARAM, "ImportState: win_size mismatch");
  BT_TRY(hipSetDevice(b->device));
  BT_TRY(hipStreamSynchronize(b->stream));
  float blk[kAnyStateFloats];
  memset(blk, 0, sizeof blk);
  memcpy(blk + kOffInTail, in->inbuf_tail, sizeof(float) * b->half);
  memcpy(blk + b->off_out, in->out_tail, sizeof(float) * b->half);
  BT_TRY(hipMemcpy(b->state + (size_t)stream * b->state_floats, blk, sizeof(float) * b->state_floats, hipMemcpyHostToDevice));
  return ASP_OK;
}

static int bt_fft_seam(const float* src, float* dst, int n, int count, int inverse, int device) {
  const bool any = n != 256 && n != 1024;
  if (!src || !dst || count <= 0 || (any && !bt_any_window_ok(n)))
    return bt_fail(ASP_ERR_PARAM, "kiss_fftr seam: bad argument");
  int rc = bt_select_device(device);
  if (rc) return rc;
  BtTables* T = nullptr;
  BtAnyTables A;
  void* any_block = nullptr;
  rc = any ? bt_build_any(n, &A, &any_block) : bt_tables(device, &T);
  if (rc) return rc;
  const size_t tb = (size_t)count * n * 4, fb = (size_t)count * (n + 2) * 4;
  float *ds = nullptr, *dd = nullptr;
  hipError_t e = hipMalloc((void**)&ds, inverse ? fb : tb);  // every failure below frees ds, dd and any_block
  if (e == hipSuccess) e = hipMalloc((void**)&dd, inverse ? tb : fb);
  if (e == hipSuccess) e = hipMemcpy(ds, src, inverse ? fb : tb, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = any ? launch_bt_fftr_any(A, ds, dd, count, inverse, nullptr) : launch_bt_fftr(n, ds, dd, count, inverse, T, nullptr);
  if (e == hipSuccess) e = hipMemcpy(dst, dd, inverse ? tb : fb, hipMemcpyDeviceToHost);
  if (ds) (void)hipFree(ds);
  if (dd) (void)hipFree(dd);
  if (any_block) (void)hipFree(any_block);
  if (e != hipSuccess) return bt_fail(ASP_ERR_HIP, "kiss_fftr seam", e);
  return ASP_OK;
}

int AspBt_kiss_fftr_batch(const float* timedata, float* freqdata, int n, int count, int device) {
  AspDeviceScope dev_scope_;
  return bt_fft_seam(timedata, freqdata, n, count, 0, device);
}
int AspBt_kiss_fftri_batch(const float* freqdata, float* timedata, int n, int count, int device) {
  AspDeviceScope dev_scope_;
  return bt_fft_seam(freqdata, timedata, n, count, 1, device);
}

// ----------------------------------------------------------------- layer 1
// audioDenoiseBlockTreshold.h:46-74 over a batch of one; hops are buffered on the host
// until a macroblock (8 hops) is complete (.c:541-575).

struct MarsBlockThreshold {
  AspBtBatch* batch;
  int32_t win_size, half_win_size, macro_size, have_nblk_time;
  float* pending;  // [macro]
  float* outbuf;   // [macro]
};

static int16_t bt_float_to_s16(float v) {  // .c:259-265
  if (v > 0) return v >= 1 ? 32767 : (int16_t)(v * 32767 + 0.5f);
  return v <= -1 ? (int16_t)(-32768) : (int16_t)(-v * (-32768) - 0.5);
}
static float bt_s16_to_float(int16_t v) {  // .c:267-271
  static const float kMaxInt16Inverse = 1.f / 32767;
  static const float kMinInt16Inverse = 1.f / (-32768);
  return v * (v > 0 ? kMaxInt16Inverse : -kMinInt16Inverse);
}

MarsBlockThreshold_t* blockThreshold_init(int32_t time_win, int32_t fs, int32_t* err) {
  AspDeviceScope dev_scope_;
  int32_t dummy;
  if (!err) err = &dummy;
  if (time_win <= 0 || fs <= 0) {
    *err = MARS_ERROR_PARAMS;
    return NULL;
  }
  int32_t win = fs / 1000 * time_win;  // .c:91-94
  if (win & 0x01) win += 1;
  if (win != 256 && win != 1024 && !bt_any_window_ok(win)) {
    fprintf(stderr,
            "blockThreshold_init: window of %d samples is not built (even, 4 .. 2048 samples, no prime factor of "
            "half the window above 32)\n", win);
    *err = MARS_ERROR_PARAMS;
    return NULL;
  }
  MarsBlockThreshold* h = (MarsBlockThreshold*)calloc(1, sizeof *h);
  if (!h || AspBtBatch_Create(&h->batch, 1, win, 0) != ASP_OK) {
    free(h);
    *err = MARS_ERROR_MEMORY;
    return NULL;
  }
  h->win_size = win;
  h->half_win_size = win / 2;
  h->macro_size = h->half_win_size * 8;
  h->pending = (float*)calloc((size_t)h->macro_size, sizeof(float));
  h->outbuf = (float*)calloc((size_t)h->macro_size, sizeof(float));
  *err = MARS_OK;
  return h;
}

int32_t blockThreshold_reset(MarsBlockThreshold_t* h) {
  AspDeviceScope dev_scope_;
  if (!h) return MARS_ERROR_PARAMS;
  h->have_nblk_time = 0;
  memset(h->outbuf, 0, sizeof(float) * (size_t)h->macro_size);
  return AspBtBatch_Reset(h->batch) == ASP_OK ? MARS_OK : MARS_ERROR_MEMORY;
}

int32_t blockThreshold_denoise_float(MarsBlockThreshold_t* h, float* in, int32_t in_len) {
  AspDeviceScope dev_scope_;
  if (!h || (in_len != h->half_win_size) || (!in)) return MARS_ERROR_PARAMS;
  memcpy(h->pending + (size_t)h->have_nblk_time * h->half_win_size, in,
         sizeof(float) * (size_t)h->half_win_size);
  h->have_nblk_time++;
  if (h->have_nblk_time != 8) return MARS_NEED_MORE_SAMPLES;
  // a failed launch is reported with the reference's own code (audioDenoiseBlockTreshold.h:9); the eight hops stay
  // pending, so the call can be repeated
  if (AspBtBatch_Denoise(h->batch, h->pending, h->outbuf, ASP_MEM_HOST) != ASP_OK) {
    h->have_nblk_time = 7;
    return MARS_ERROR_MEMORY;
  }
  h->have_nblk_time = 0;
  return MARS_CAN_OUTPUT;
}

int32_t blockThreshold_denoise_int16(MarsBlockThreshold_t* h, int16_t* in, int32_t in_len) {
  AspDeviceScope dev_scope_;
  if (!h || (in_len != h->half_win_size) || (!in)) return MARS_ERROR_PARAMS;
  std::vector<float> tmp((size_t)in_len);
  for (int32_t i = 0; i < in_len; i++) tmp[i] = bt_s16_to_float(in[i]);
  return blockThreshold_denoise_float(h, tmp.data(), in_len);
}

int32_t blockThreshold_output_float(MarsBlockThreshold_t* h, float* out, int32_t out_len) {
  AspDeviceScope dev_scope_;
  if (out_len < h->macro_size) return 0;
  memcpy(out, h->outbuf, sizeof(float) * (size_t)h->macro_size);
  return h->macro_size;
}

int32_t blockThreshold_output_int16(MarsBlockThreshold_t* h, int16_t* out, int32_t out_len) {
  AspDeviceScope dev_scope_;
  if (out_len < h->macro_size) return 0;
  for (int32_t i = 0; i < h->macro_size; i++) out[i] = bt_float_to_s16(h->outbuf[i]);
  return h->macro_size;
}

int32_t blockThreshold_flush_float(MarsBlockThreshold_t* h, float* out, int32_t out_len) {
  AspDeviceScope dev_scope_;
  const int32_t out_size = h->have_nblk_time * h->half_win_size;
  if (out_len < out_size) return -1;
  // with no pending hop the reference still shifts the overlap tail out of outbuf and clears it
  // (.c:656-659): the device call with hops = 0 does the same to the stream's tail
  float none = 0.f;
  if (AspBtBatch_Flush(h->batch, h->pending, h->have_nblk_time, out_size ? out : &none, ASP_MEM_HOST) != ASP_OK)
    return -1;
  return out_size;
}

int32_t blockThreshold_flush_int16(MarsBlockThreshold_t* h, int16_t* out, int32_t out_len) {
  AspDeviceScope dev_scope_;
  const int32_t out_size = h->have_nblk_time * h->half_win_size;
  if (out_len < out_size) return -1;
  std::vector<float> tmp((size_t)out_size + 1);
  if (AspBtBatch_Flush(h->batch, h->pending, h->have_nblk_time, tmp.data(), ASP_MEM_HOST) != ASP_OK)
    return -1;
  for (int32_t i = 0; i < out_size; i++) out[i] = bt_float_to_s16(tmp[i]);
  return out_size;
}

void blockThreshold_free(MarsBlockThreshold_t* h) {
  AspDeviceScope dev_scope_;
  if (!h) return;
  AspBtBatch_Free(h->batch);
  free(h->pending);
  free(h->outbuf);
  free(h);
}

int32_t blockThreshold_max_output(const MarsBlockThreshold_t* h) { return h->macro_size; }
int32_t blockThreshold_samples_per_time(const MarsBlockThreshold_t* h) { return h->half_win_size; }

}  // extern "C"
