// aec_api.hip -- host side of include/asp_aec.h: constant tables, the batch handle with the
// reference's integer control plane (echo_cancellation.c:196-409,594-742,816-867;
// aec_core.c:1618-1778; ring positions of common_audio/ring_buffer.c) and the reference's
// per-stream WebRtcAec_* symbols as a batch of one.  The control plane runs once per call for
// the whole batch and turns every call into launch descriptors (aec_layout.h: FarOps /
// ProcOps); all sample and spectrum data stays in HBM.  No CPU fallback.
#include <hip/hip_runtime.h>

#include "device_scope.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "aec_layout.h"
#include "asp_aec.h"
#include "asp_ns.h"

using namespace aspaec;

namespace aspaec {
hipError_t launch_aec_farend(float* state, float* far_ring, const AecTables* T, const float* farend,
                             int num_streams, const FarOps& ops, hipStream_t s, int stream0 = 0,
                             int stream_end = -1, int num_part = kNumPartNormal);
hipError_t launch_aec_process(float* state, float* far_ring, const AecTables* T,
                              const float* nearend, float* out, int num_streams, int nrOfSamples,
                              const ProcOps& ops, const float* farend, const FarOps& fops,
                              const float* near_high, float* out_high, float* metrics,
                              hipStream_t s, unsigned long long* stamps = nullptr, int stream0 = 0,
                              int stream_end = -1, int num_part = kNumPartNormal, float* spectra = nullptr,
                              const DelayBlock* dblocks = nullptr);
hipError_t launch_aec_process_flow(float* state, float* far_ring, const AecTables* T, int num_streams, int nrOfSamples,
                                   const AecFlowStep* descs, int steps, unsigned* seq, unsigned* abort_w, unsigned want,
                                   int num_part, hipStream_t s, DelayBlock* est, unsigned* bits, bool agn);
hipError_t launch_aec_process_agn(float* state, float* far_ring, const AecTables* T, const float* nearend, float* out,
                                  int num_streams, int nrOfSamples, const ProcOps& ops, const float* farend, const FarOps& fops,
                                  DelayBlock* dblocks, const AgnOps& agn, hipStream_t s, int stream0, int stream_end,
                                  int num_part);
hipError_t launch_aec_delay_bits(DelayBlock* blocks, const unsigned* bits, int num_streams, int npending, int logging,
                                 hipStream_t s);
hipError_t launch_aec_farend_v(float* state, float* far_ring, const AecTables* T, const float* farend, int num_streams,
                               const FarOps* vfar, int nrOfSamples, int num_part, hipStream_t s);
hipError_t launch_aec_process_v(float* state, float* far_ring, const AecTables* T, const float* nearend, float* out,
                                int num_streams, int nrOfSamples, const AecStreamStep* vdesc, int num_part, hipStream_t s);
hipError_t launch_aec_delay(DelayBlock* blocks, const float* spectra, int num_streams, const DelayOps& ops,
                            hipStream_t s);
hipError_t launch_aec_resample(float* rs_buffer, const float* farend, float* out, int num_streams, int size, int size_out,
                               float be, float position, hipStream_t s);
hipError_t launch_aec_rdft128(const float* src, float* dst, int isgn, int count, const AecTables* T,
                              hipStream_t s);
}  // namespace aspaec

namespace {

thread_local char g_aec_err[512] = "";
int aec_fail(int code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess)
    snprintf(g_aec_err, sizeof g_aec_err, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(g_aec_err, sizeof g_aec_err, "%s", what);
  fprintf(stderr, "asp_aec: %s\n", g_aec_err);
  return code;
}
#define AEC_TRY(expr)                                             \
  do {                                                            \
    hipError_t e_ = (expr);                                       \
    if (e_ != hipSuccess) return aec_fail(ASP_ERR_HIP, #expr, e_); \
  } while (0)

// ------------------------------------------------------------------ tables
// NOTE: C++ translation unit; every libm call casts to double explicitly so the arithmetic is
// that of the reference's C (see ns_api.hip).
unsigned bitrev(unsigned x, int bits) {
  unsigned r = 0;
  for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1u) << (bits - 1 - b);
  return r;
}

float via_text(double v, int decimals) {  // a table frozen as '%.<d>f' text and re-read as float
  char buf[48];
  snprintf(buf, sizeof buf, "%.*f", decimals, v);
  return strtof(buf, nullptr);
}

void build_tables(AecTables* T) {
  memset(T, 0, sizeof *T);
  // Ooura makewt / makect for n = 128, the code the reference's frozen rdft_w came from
  // (aec_rdft.c:29-49; utility/fft4g.c:642-690 is the same code)
  float tmp[32];
  const int nw = 32, nwh = 16, nc = 32, nch = 16;
  float delta = (float)atan((double)1.0f) / nwh;
  tmp[0] = 1;
  tmp[1] = 0;
  tmp[nwh] = (float)cos((double)(delta * nwh));
  tmp[nwh + 1] = tmp[nwh];
  for (int j = 2; j < nwh; j += 2) {
    const float x = (float)cos((double)(delta * j));
    const float y = (float)sin((double)(delta * j));
    tmp[j] = x;
    tmp[j + 1] = y;
    tmp[nw - j] = y;
    tmp[nw - j + 1] = x;
  }
  for (int j = 0; j < 16; ++j) {
    const unsigned r = bitrev((unsigned)j, 4);
    T->w[2 * j] = tmp[2 * r];
    T->w[2 * j + 1] = tmp[2 * r + 1];
  }
  delta = (float)atan((double)1.0f) / nch;
  T->w[32] = (float)cos((double)(delta * nch));
  T->w[32 + nch] = 0.5f * T->w[32];
  for (int j = 1; j < nch; j++) {
    T->w[32 + j] = 0.5f * (float)cos((double)(delta * j));
    T->w[32 + nc - j] = 0.5f * (float)sin((double)(delta * j));
  }
  {
    // the frozen text differs from the evaluation above by one unit in the last place at eight
    // entries (data of the reference, aec_rdft.c:32-49)
    static const signed char nudge[8][2] = {{4, 1}, {7, 1}, {20, 1}, {27, 1},
                                            {40, 1}, {41, -1}, {42, 1}, {47, 1}};
    for (int k = 0; k < 8; ++k) {
      int32_t bits;
      memcpy(&bits, &T->w[nudge[k][0]], sizeof bits);
      bits += nudge[k][1];
      memcpy(&T->w[nudge[k][0]], &bits, sizeof bits);
    }
  }
  for (int k1 = 0; k1 < 16; k1 += 2) {  // rdft_wk3ri_first / _second (aec_rdft.c:50-61)
    const int k2 = 2 * k1;
    const float wk2r = T->w[k1], wk2i = T->w[k1 + 1];
    float wk1r = T->w[k2], wk1i = T->w[k2 + 1];
    T->wk3a[k1] = wk1r - 2 * wk2i * wk1i;
    T->wk3a[k1 + 1] = 2 * wk2i * wk1r - wk1i;
    wk1r = T->w[k2 + 2];
    wk1i = T->w[k2 + 3];
    T->wk3b[k1] = wk1r - 2 * wk2r * wk1i;
    T->wk3b[k1 + 1] = 2 * wk2r * wk1r - wk1i;
  }
  for (int k = 0; k <= 64; ++k) {  // aec_core.c:50-98
    T->hann[k] = via_text(sin(M_PI * (double)k / 128.0), 14);
    T->weight[k] = k == 0 ? 0.f : via_text(0.3 * sqrt((double)(k - 1) / 63.0) + 0.1, 4);
    T->odrive[k] = via_text(sqrt((double)k / 64.0) + 1.0, 4);
  }
  for (int j = 0; j < 64; ++j) T->exp2_64[j] = exp2((double)j / 64.0);
  uint32_t a = 1, c = 0;  // WebRtcSpl_RandU jump-ahead (randomization_functions.c:93-100)
  for (int k = 0; k < 64; ++k) {
    c = c * 69069u + 1u;
    a = a * 69069u;
    T->lcg_a[k] = a;
    T->lcg_c[k] = c;
  }
}

// ------------------------------------------------------------ ring positions
struct RingPos {  // common_audio/ring_buffer.c:26-33 without the data
  int read, write, wrap, count;
};
void rp_init(RingPos* r, int count) {
  r->read = 0;
  r->write = 0;
  r->wrap = 0;
  r->count = count;
}
int rp_avail_read(const RingPos* r) {
  return r->wrap == 0 ? r->write - r->read : r->count - r->read + r->write;
}
int rp_avail_write(const RingPos* r) { return r->count - rp_avail_read(r); }
int rp_move_read(RingPos* r, int n) {  // WebRtc_MoveReadPtr, ring_buffer.c:195-228
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
// WebRtc_WriteBuffer positions (ring_buffer.c:161-192): returns the elements written; *start is
// the position the first element lands on (the device adds i modulo count).
int rp_write(RingPos* r, int n, int* start) {
  const int free_elements = rp_avail_write(r);
  const int write_elements = free_elements < n ? free_elements : n;
  int m = write_elements;
  const int margin = r->count - r->write;
  *start = r->write;
  if (write_elements > margin) {
    r->write = 0;
    m -= margin;
    r->wrap = 1;
  }
  r->write += m;
  return write_elements;
}
// WebRtc_ReadBuffer positions (ring_buffer.c:112-158)
int rp_read(RingPos* r, int n, int* start) {
  const int readable = rp_avail_read(r);
  const int read_elements = readable < n ? readable : n;
  *start = r->read;
  rp_move_read(r, read_elements);
  return read_elements;
}

const int kInitCheck = 42;           // echo_cancellation.c:59
const int kMaxTrustedDelayMs = 500;  // :53
const int kMaxBufSizeStart = 62;     // :57
const int kSampMsNb = 8;             // :58

}  // namespace

// The control plane of ONE stream -- every integer the reference keeps per handle (Aec, echo_cancellation_internal.h:
// 17-65; the integer part of AecCore, aec_core_internal.h:52-164; ring positions, ring_buffer.c:26-33) -- as plain
// data.  A batch fed in lock-step has one of these (its base); once a caller drives streams apart (per-stream
// reported delays, a stream re-initialised: AspAecBatch_ProcessV / _InitStream) it has one per stream (`per`), and
// the control functions below run once per stream on a copy swapped into the base.
struct AecCtl {
  // WebRtcAec_enable_delay_correction (aec_core.c:1876-1881): the extended filter, 32 partitions instead of 12
  int extended = 0, num_part = kNumPartNormal;
  int delay_logging = 0, reported_delay_enabled = 1;
  bool agn_synced = false;
  int nevents = 0, ev_samples[kMaxFarEvents] = {}, ev_parts[kMaxFarEvents] = {};
  // skew compensation (set_config skewMode; echo_cancellation.c:304-313, 614-645, aec_resampler.c): the skew estimate
  // and the resampler's position are functions of the calls' skew arguments alone -- one copy for the batch, on the
  // host; the resampler's sample buffer is per stream, on the device
  int skewFrCtr = 0, resample = 0;
  float skew = 0.f, sampFactor = 1.f;
  float rs_position = 0.f;
  int rs_skewDataIndex = 0;
  float rs_skewEstimate = 0.f;
  // Aec (echo_cancellation_internal.h:17-65)
  int sampFreq = 0, scSampFreq = 0, splitSampFreq = 0, rate_factor = 0, initFlag = 0, lastError = 0;
  int farend_started = 0, skewMode = 0;
  int bufSizeStart = 0, knownDelay = 0, timeForDelayChange = 0, startup_phase = 0, checkBuffSize = 0, sum = 0;
  int16_t counter = 0, firstVal = 0, checkBufSizeCtr = 0, msInSndCardBuf = 0, filtDelay = 0, lastDelayDiff = 0;
  // integer part of AecCore (aec_core_internal.h:52-164)
  int system_delay = 0, core_knownDelay = 0, mult = 0, nlp_mode = 1;
  float normal_mu = 0.f, normal_error_threshold = 0.f;
  int xf_pos = 0, xfw_head = 0, blocks_processed = 0;
  RingPos pre_pos{}, far_pos{}, near_pos{}, out_pos{};
  // a BufferFarend whose device work is deferred into the next Process launch (Run / TimedSteps)
  bool far_pending = false;
  FarOps far_ops{};
  const float* far_src = nullptr;
  // 32 kHz: one high band (aec_core.c:1032-1067)
  int num_high = 0;
  int metricsMode = 0;
};

struct AspAecBatch : AecCtl {
  int S = 0, device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  float* state = nullptr;     // [S][AecRows(num_part).state_dwords]
  // delay estimation (set_config delay_logging) and the delay-agnostic mode (WebRtcAec_enable_reported_delay 0):
  // per-stream estimator blocks, the power spectra the process kernel leaves for them, and -- once the agnostic mode
  // has taken over a stream's far-buffer read side (agn_synced) -- the BufferFarend calls the device has yet to replay
  DelayBlock* dblocks = nullptr;  // [S]
  float* spectra = nullptr;       // [S][kSpecBlocks][kSpecDwords]
  int rs_skewData[kSkewEstimateFrames] = {};
  float* rs_buffer = nullptr;  // [S][kResamplerBufferSize]
  float* stage_rs = nullptr;   // [S][kResamplerBufferSize]: the resampled far frame of the call in flight
  float* far_ring = nullptr;  // [kFarSlots][S][kFarSlotDwords]
  AecTables* tables = nullptr;
  float *stage_far = nullptr, *stage_near = nullptr, *stage_out = nullptr;  // [S][160]
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // two launch chains over the two halves of the batch (the K-step path, AspAecBatch_TimedSteps): streams are
  // independent, so half B's step k may still run while half A's step k + 1 starts -- the load burst of one
  // overlaps the transforms of the other
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool dual = false;  // set only inside the K-step path
  // Hand-off build of the multi-frame entry points (Run on device buffers, TimedSteps; aec_kernels.hip,
  // AecFlowArgs): while flow_rec is set the Process launches are not issued one by one but recorded as
  // descriptors; up to kAecFlowMaxSteps of them go into ONE launch (grid y = step), in which a per-stream step
  // counter in memory orders a stream's consecutive steps.  -1 = default (on), 0 = off, 1 = on.
  int flow = -1;
  bool flow_rec = false;
  int flow_n = 0, flow_nr = 0;               // descriptors recorded so far; samples per call of the recording
  int flow_slot = 0;                         // ring of descriptor arrays: host (pinned) and device copies
  unsigned char* flow_host = nullptr;        // [kAecFlowSlots][kAecFlowMaxSteps] AecFlowStep, or AecFlowStepAgn in the delay-agnostic mode
  unsigned char* flow_dev = nullptr;
  bool flow_agn = false;                     // the recording's element type
  hipEvent_t flow_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // slot i's launch has consumed its descriptors
  unsigned* flow_seq = nullptr;              // [S] completed hand-off steps per stream (== flow_count between calls)
  unsigned* flow_abort = nullptr;            // 16 B: word 0 != 0 after a wait timed out
  unsigned flow_count = 0;
  bool flow_unchecked = false;
  unsigned* flow_bits = nullptr;             // delay logging: [S][kFlowBitsBlocks][2] binary spectra of the recorded steps' blocks
  int flow_blocks = 0;                       // blocks of the recording so far (their spectra wait for aec_delay_bits_kernel)
  // Per-stream control (AspAecBatch_ProcessV / _InitStream): one control plane per stream on the host, the launch
  // descriptors of a call recorded per stream and read by the kernels from device memory
  std::vector<AecCtl> per;                   // empty while the batch runs in lock-step
  bool vrec = false;                         // a per-stream control step is recording (vstream)
  int vstream = 0;
  AecStreamStep* vproc_host = nullptr;       // [S], pinned: this call's Process descriptor of every stream
  AecStreamStep* vproc_dev = nullptr;
  FarOps* vfar_host = nullptr;               // [S], pinned: this call's far-end work of every stream
  FarOps* vfar_dev = nullptr;
  int vfar_recorded = 0;
  hipEvent_t vev = nullptr;                  // the last per-stream launch has read its descriptors
  unsigned long long* debug_stamps = nullptr;  // diagnostic only (AspAecBatch_DebugStamps)
  // control-plane-only handle (AspAecBatch_CreateControlOnly): no device, nothing is launched
  bool sim = false;
  float *stage_near_h = nullptr, *stage_out_h = nullptr;  // [S][160]
  const float* cur_near_high = nullptr;  // device pointers of the Process call in flight
  float* cur_out_high = nullptr;
  // echo metrics (aec_core.c:548-770): [S][kMetDwords], updated by the kernel when metricsMode is on
  float* metrics = nullptr;
};


namespace {
// EstimateSkew (aec_resampler.c:143-217): least-squares slope of the cumulated, outlier-cleaned skew reports
int estimate_skew(const int* rawSkew, int size, int deviceSampleRateHz, float* skewEst) {
  const int absLimitOuter = (int)(0.04f * deviceSampleRateHz);
  const int absLimitInner = (int)(0.0025f * deviceSampleRateHz);
  int n = 0;
  float rawAvg = 0, err = 0, rawAbsDev = 0, cumSum = 0, x = 0, x2 = 0, y = 0, xy = 0, xAvg = 0, denom = 0, skew = 0;
  *skewEst = 0;
  for (int i = 0; i < size; i++) {
    if ((rawSkew[i] < absLimitOuter && rawSkew[i] > -absLimitOuter)) {
      n++;
      rawAvg += rawSkew[i];
    }
  }
  if (n == 0) return -1;
  rawAvg /= n;
  for (int i = 0; i < size; i++) {
    if ((rawSkew[i] < absLimitOuter && rawSkew[i] > -absLimitOuter)) {
      err = rawSkew[i] - rawAvg;
      rawAbsDev += err >= 0 ? err : -err;
    }
  }
  rawAbsDev /= n;
  const int upperLimit = (int)(rawAvg + 5 * rawAbsDev + 1);
  const int lowerLimit = (int)(rawAvg - 5 * rawAbsDev - 1);
  n = 0;
  for (int i = 0; i < size; i++) {
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
}  // namespace

namespace {
constexpr int kAecFlowMaxSteps = 64, kAecFlowSlots = 4;
static_assert(kFlowBitsBlocks >= 4 * kAecFlowMaxSteps, "a step has at most four blocks");
constexpr size_t kAecFlowElem = sizeof(AecFlowStepAgn);  // a slot holds kAecFlowMaxSteps elements of either type

bool aec_flow_default() {
  const char* e = getenv("ASP_AEC_FLOW");
  return !(e && e[0] == '0');
}
// The delay-agnostic mode in one launch per call (aec_process_agn_kernel): one band, no echo metrics, the whole batch on
// one control plane.  ASP_AEC_AGN_FUSED=0 keeps the launch-per-sub-frame form (the A / B switch).
bool agn_fused(const AspAecBatch* b) {
  static const int env = [] {
    const char* e = getenv("ASP_AEC_AGN_FUSED");
    return e ? atoi(e) : 1;
  }();
  return env != 0 && !b->sim && !b->vrec && b->num_high == 0 && !b->metricsMode && b->debug_stamps == nullptr;
}

// The hand-off build serves one band without skew compensation and echo metrics: the plain configuration, delay
// logging (the estimator's launch follows each process launch) and the fused delay-agnostic mode.
bool aec_flow_applies(const AspAecBatch* b, int steps) {
  const bool on = b->flow < 0 ? aec_flow_default() : b->flow != 0;
  if (!b->per.empty()) return false;  // per-stream control: one far-end and one Process launch per call
  return on && steps >= 2 && !b->sim && b->num_high == 0 && !b->metricsMode &&
         (b->reported_delay_enabled || agn_fused(b)) && !b->skewMode && b->debug_stamps == nullptr;
}
int aec_flow_resources(AspAecBatch* b) {
  if (b->delay_logging && b->flow_bits == nullptr)
    AEC_TRY(hipMalloc((void**)&b->flow_bits, (size_t)b->S * kFlowBitsBlocks * 2 * sizeof(unsigned)));
  if (b->flow_seq) return 0;
  AEC_TRY(hipMalloc((void**)&b->flow_seq, (size_t)b->S * sizeof(unsigned)));
  AEC_TRY(hipMalloc((void**)&b->flow_abort, 16));
  AEC_TRY(hipMemsetAsync(b->flow_seq, 0, (size_t)b->S * sizeof(unsigned), b->stream));
  AEC_TRY(hipMemsetAsync(b->flow_abort, 0, 16, b->stream));
  AEC_TRY(hipHostMalloc((void**)&b->flow_host, kAecFlowElem * kAecFlowSlots * kAecFlowMaxSteps, hipHostMallocDefault));
  AEC_TRY(hipMalloc((void**)&b->flow_dev, kAecFlowElem * kAecFlowSlots * kAecFlowMaxSteps));
  for (int i = 0; i < kAecFlowSlots; ++i) AEC_TRY(hipEventCreateWithFlags(&b->flow_ev[i], hipEventDisableTiming));
  b->flow_count = 0;
  b->flow_slot = 0;
  return 0;
}
// issue the recorded steps as one launch
int aec_flow_flush(AspAecBatch* b) {
  if (b->flow_n == 0) return 0;
  const int slot = b->flow_slot;
  unsigned char* h = b->flow_host + (size_t)slot * kAecFlowMaxSteps * kAecFlowElem;
  unsigned char* d = b->flow_dev + (size_t)slot * kAecFlowMaxSteps * kAecFlowElem;
  const int n = b->flow_n;
  b->flow_n = 0;
  const bool agn = b->flow_agn;
  AEC_TRY(hipMemcpyAsync(d, h, (agn ? sizeof(AecFlowStepAgn) : sizeof(AecFlowStep)) * n, hipMemcpyHostToDevice, b->stream));
  const int blocks = b->flow_blocks;  // > 0: delay logging is on
  b->flow_blocks = 0;
  AEC_TRY(launch_aec_process_flow(b->state, b->far_ring, b->tables, b->S, b->flow_nr, reinterpret_cast<const AecFlowStep*>(d), n,
                                  b->flow_seq, b->flow_abort, b->flow_count, b->num_part, b->stream,
                                  (agn || blocks > 0) ? b->dblocks : nullptr, blocks > 0 ? b->flow_bits : nullptr, agn));
  // the estimator's share of these steps (aec_core.c:1191-1203): one launch for all their blocks, from the binary
  // spectra the process kernel left
  if (blocks > 0) AEC_TRY(launch_aec_delay_bits(b->dblocks, b->flow_bits, b->S, blocks, 1, b->stream));
  AEC_TRY(hipEventRecord(b->flow_ev[slot], b->stream));
  b->flow_count += (unsigned)n;
  b->flow_unchecked = true;
  b->flow_slot = (slot + 1) % kAecFlowSlots;
  // the next slot's previous launch must have read its descriptors before they are overwritten
  AEC_TRY(hipEventSynchronize(b->flow_ev[b->flow_slot]));
  return 0;
}
int aec_flow_record(AspAecBatch* b, const float* near_dev, float* out_dev, int n, const ProcOps& ops, const float* far_src,
                    const FarOps& fops, const AgnOps* agn = nullptr) {
  if (b->flow_n > 0 && (b->flow_nr != n || b->flow_agn != (agn != nullptr))) {  // a launch carries calls of one length and one kind
    const int rc = aec_flow_flush(b);
    if (rc != 0) return rc;
  }
  b->flow_nr = n;
  b->flow_agn = agn != nullptr;
  unsigned char* slot_base = b->flow_host + (size_t)b->flow_slot * kAecFlowMaxSteps * kAecFlowElem;
  const int idx = b->flow_n++;
  AecFlowStepAgn* sa = agn ? reinterpret_cast<AecFlowStepAgn*>(slot_base) + idx : nullptr;
  AecFlowStep& st = agn ? sa->step : reinterpret_cast<AecFlowStep*>(slot_base)[idx];
  if (agn) sa->agn = *agn;
  memset(&st, 0, sizeof st);
  st.ops = ops;
  st.fops = fops;
  st.farend = far_src;
  st.nearend = near_dev;
  st.out = out_dev;
  st.spec_base = -1;
  if (b->delay_logging && agn == nullptr) {  // (the delay-agnostic mode runs the estimator in the process wave)
    st.spec_base = b->flow_blocks;
    for (int j = 0; j < ops.nsub; ++j) b->flow_blocks += ops.sub[j].nblocks;  // <= 4 per step, kFlowBitsBlocks = 4 * kAecFlowMaxSteps
  }
  if (b->flow_n == kAecFlowMaxSteps) return aec_flow_flush(b);
  return 0;
}
// after the batch's stream has been synchronised: did a hand-off wait time out?
int aec_flow_check(AspAecBatch* b) {
  if (!b->flow_unchecked) return 0;
  b->flow_unchecked = false;
  unsigned a = 0;
  AEC_TRY(hipMemcpy(&a, b->flow_abort, sizeof a, hipMemcpyDeviceToHost));
  if (a == 0) return 0;
  std::vector<unsigned> seq((size_t)b->S, b->flow_count);
  AEC_TRY(hipMemcpy(b->flow_seq, seq.data(), seq.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  AEC_TRY(hipMemset(b->flow_abort, 0, 16));
  return aec_fail(ASP_ERR_HIP, "AEC hand-off wait timed out: frame steps were skipped, re-initialise the batch");
}

// every device launch of a batch goes through these two: one launch, or one per half on the two chains
hipError_t batch_launch_farend(AspAecBatch* b, const float* far_dev, const FarOps& ops) {
  if (b->vrec) {  // per-stream control: this stream's far-end work of the call (one descriptor: nothing is deferred)
    if (b->vfar_recorded++ != 0) return hipErrorUnknown;
    b->vfar_host[b->vstream] = ops;
    return hipSuccess;
  }
  if (b->flow_rec && aec_flow_flush(b) != 0) return hipErrorUnknown;  // a launch of its own: after the recorded steps
  if (!b->dual) return launch_aec_farend(b->state, b->far_ring, b->tables, far_dev, b->S, ops, b->stream, 0, -1, b->num_part);
  const int half = ((b->S / 2 + 3) / 4) * 4;
  hipError_t e = launch_aec_farend(b->state, b->far_ring, b->tables, far_dev, b->S, ops, b->stream, 0, half, b->num_part);
  if (e == hipSuccess)
    e = launch_aec_farend(b->state, b->far_ring, b->tables, far_dev, b->S, ops, b->side, half, b->S, b->num_part);
  return e;
}
hipError_t batch_launch_process(AspAecBatch* b, const float* near_dev, float* out_dev, int n, const ProcOps& ops,
                                const float* far_src, const FarOps& fops, const float* near_high, float* out_high,
                                float* metrics, unsigned long long* stamps) {
  if (b->vrec) {  // per-stream control: this stream's Process descriptor
    if (far_src != nullptr) return hipErrorUnknown;
    b->vproc_host[b->vstream].mode = 1;
    b->vproc_host[b->vstream].ops = ops;
    return hipSuccess;
  }
  if (b->flow_rec) return aec_flow_record(b, near_dev, out_dev, n, ops, far_src, fops) == 0 ? hipSuccess : hipErrorUnknown;
  if (!b->dual)
    return launch_aec_process(b->state, b->far_ring, b->tables, near_dev, out_dev, b->S, n, ops, far_src, fops, near_high,
                              out_high, metrics, b->stream, stamps, 0, -1, b->num_part, b->spectra, b->dblocks);
  const int half = ((b->S / 2 + 3) / 4) * 4;
  hipError_t e = launch_aec_process(b->state, b->far_ring, b->tables, near_dev, out_dev, b->S, n, ops, far_src, fops,
                                    near_high, out_high, metrics, b->stream, stamps, 0, half, b->num_part, b->spectra, b->dblocks);
  if (e == hipSuccess)
    e = launch_aec_process(b->state, b->far_ring, b->tables, near_dev, out_dev, b->S, n, ops, far_src, fops, near_high,
                           out_high, metrics, b->side, nullptr, half, b->S, b->num_part, b->spectra, b->dblocks);
  return e;
}
}  // namespace

namespace {

int far_move_read(AspAecBatch* b, int elements) {  // WebRtcAec_MoveFarReadPtr, aec_core.c:1637-1645
  const int moved = rp_move_read(&b->far_pos, elements);
  b->system_delay -= moved * kPartLen;
  return moved;
}

// WebRtc_InitDelayEstimatorFarend / WebRtc_InitDelayEstimator and the AecCore fields around them (aec_core.c:1502-1516,
// delay_estimator_wrapper.c:175-191, 305-322, delay_estimator.c:303-307, 476-498) for every stream.  The lookahead
// is not part of Init: a stream keeps what Create set or the agnostic mode's soft resets left (as the reference does).
void init_delay_state(AspAecDelayState* d, int lookahead, int allowed_offset) {
  memset(d, 0, sizeof *d);
  d->lookahead = lookahead;
  d->allowed_offset = allowed_offset;
  for (int i = 0; i <= ASP_AEC_DELAY_HISTORY; ++i) d->mean_bit_counts[i] = (20 << 9);
  d->minimum_probability = 32 << 9;
  d->last_delay_probability = 32 << 9;
  d->last_delay = -2;
  d->last_candidate_delay = -2;
  d->compare_delay = ASP_AEC_DELAY_HISTORY;
  d->previous_delay = -2;
  d->shift_offset = 5;  // kInitialShiftOffset, aec_core.c:102
}
int init_delay_device(AspAecBatch* b) {
  std::vector<DelayBlock> all((size_t)b->S);
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  AEC_TRY(hipMemcpy(all.data(), b->dblocks, all.size() * sizeof(DelayBlock), hipMemcpyDeviceToHost));
  for (auto& d : all) {
    const int lookahead = d.s.lookahead;
    memset(&d, 0, sizeof d);
    init_delay_state(&d.s, lookahead, kNumPartNormal / 2);  // WebRtc_set_allowed_offset(num_partitions / 2), aec_core.c:1529
  }
  AEC_TRY(hipMemcpy(b->dblocks, all.data(), all.size() * sizeof(DelayBlock), hipMemcpyHostToDevice));
  return 0;
}

// InitMetrics (aec_core.c:548-583) for every stream, ordered on the batch's stream.
int init_metrics_device(AspAecBatch* b) {
  AspAecMetricsState m;
  memset(&m, 0, sizeof m);
  AspAecPowerLevel* lv[4] = {&m.farlevel, &m.nearlevel, &m.linoutlevel, &m.nlpoutlevel};
  for (AspAecPowerLevel* l : lv) l->minlevel = 1E17f;
  AspAecStats* st[4] = {&m.erl, &m.erle, &m.aNlp, &m.rerl};
  for (AspAecStats* s : st) {
    s->instant = s->average = s->max = s->himean = -100;  // kOffsetLevel
    s->min = 100;
  }
  static_assert(sizeof(AspAecMetricsState) == 65 * 4 && sizeof(AspAecMetricsState) <= kMetDwords * 4, "metrics image");
  std::vector<float> all((size_t)b->S * kMetDwords, 0.f);
  for (int s = 0; s < b->S; ++s) memcpy(all.data() + (size_t)s * kMetDwords, &m, sizeof m);
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  AEC_TRY(hipMemcpy(b->metrics, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int state_dwords(const AspAecBatch* b) { return AecRows(b->num_part).state_dwords; }
size_t state_bytes(const AspAecBatch* b) { return (size_t)b->S * state_dwords(b) * sizeof(float); }
size_t far_bytes(const AspAecBatch* b) {
  return (size_t)kFarSlots * b->S * kFarSlotDwords * sizeof(float);
}

// ---- AspAecState <-> device block
void pack_stream(const AspAecBatch* b, const AspAecState* s, float* blk) {
  const AecRows rw(b->num_part);
  const int kNumPart = rw.NP, R_XF_IM = rw.R_XF_IM, R_WF_RE = rw.R_WF_RE, R_WF_IM = rw.R_WF_IM, R_XFW = rw.R_XFW;
  memset(blk, 0, rw.state_dwords * sizeof(float));
  float* rows = blk + kOffRows;
  auto put_row = [&](int r, const float* src) {
    memcpy(rows + r * kRowS, src, 64 * sizeof(float));
    blk[kOffC64 + r] = src[64];
  };
  put_row(R_XPOW, s->xPow);
  put_row(R_DPOW, s->dPow);
  put_row(R_DMINPOW, s->dMinPow);
  put_row(R_DINITMINPOW, s->dInitMinPow);
  put_row(R_SX, s->sx);
  put_row(R_SD, s->sd);
  put_row(R_SE, s->se);
  {
    float tmp[4][65];
    for (int i = 0; i < 65; ++i) {
      tmp[0][i] = s->sde[i][0];
      tmp[1][i] = s->sde[i][1];
      tmp[2][i] = s->sxd[i][0];
      tmp[3][i] = s->sxd[i][1];
    }
    put_row(R_SDE_RE, tmp[0]);
    put_row(R_SDE_IM, tmp[1]);
    put_row(R_SXD_RE, tmp[2]);
    put_row(R_SXD_IM, tmp[3]);
  }
  for (int a = 0; a < kNumPart; ++a) {  // logical age a: canonical (c + a) % NP -> physical (h + a) % NP
    const int pc = (s->xfBufBlockPos + a) % kNumPart, ph = (b->xf_pos + a) % kNumPart;
    put_row(R_XF_RE + ph, s->xfBuf[0] + pc * 65);
    put_row(R_XF_IM + ph, s->xfBuf[1] + pc * 65);
  }
  for (int i = 0; i < kNumPart; ++i) {
    put_row(R_WF_RE + i, s->wfBuf[0] + i * 65);
    put_row(R_WF_IM + i, s->wfBuf[1] + i * 65);
  }
  {
    // canonical partition a >= 1 holds the block of age a - 1 (aec_core.c:1079-1081); physical
    // age g sits at (xfw_head + g) % 32: all 32 blocks of history whatever the filter length, as the
    // reference shifts them (aec_core.c:1079-1081)
    const float* raw = &s->xfwBuf[0][0];
    for (int a = 1; a < kNumPartMax; ++a) {
      const int ph = (b->xfw_head + a - 1) % kNumPartMax;
      put_row(R_XFW + 2 * ph, raw + a * 130);
      put_row(R_XFW + 2 * ph + 1, raw + a * 130 + 65);
    }
  }
  memcpy(blk + kOffDBuf, s->dBuf, 64 * sizeof(float));
  memcpy(blk + kOffEBuf, s->eBuf, 64 * sizeof(float));
  memcpy(blk + kOffOutBuf, s->outBuf, 64 * sizeof(float));
  memcpy(blk + kOffDBufH, s->dBufH, 64 * sizeof(float));
  float* sc = blk + kOffScalars;
  int32_t* sci = reinterpret_cast<int32_t*>(sc);
  sc[S_HNLFBMIN] = s->hNlFbMin;
  sc[S_HNLFBLOCALMIN] = s->hNlFbLocalMin;
  sc[S_HNLXDAVGMIN] = s->hNlXdAvgMin;
  sc[S_OVERDRIVE] = s->overDrive;
  sc[S_OVERDRIVESM] = s->overDriveSm;
  sci[S_HNLNEWMIN] = s->hNlNewMin;
  sci[S_HNLMINCTR] = s->hNlMinCtr;
  sci[S_DELAYIDX] = s->delayIdx;
  sci[S_STNEARSTATE] = s->stNearState;
  sci[S_ECHOSTATE] = s->echoState;
  sci[S_DIVERGESTATE] = s->divergeState;
  sci[S_NOISEESTCTR] = s->noiseEstCtr;
  sci[S_DELAYESTCTR] = s->delayEstCtr;
  reinterpret_cast<uint32_t*>(sc)[S_SEED] = s->seed;
}

void unpack_stream(const AspAecBatch* b, const float* blk, AspAecState* s) {
  const AecRows rw(b->num_part);
  const int kNumPart = rw.NP, R_XF_IM = rw.R_XF_IM, R_WF_RE = rw.R_WF_RE, R_WF_IM = rw.R_WF_IM, R_XFW = rw.R_XFW;
  memset(s, 0, sizeof *s);
  const float* rows = blk + kOffRows;
  auto get_row = [&](int r, float* dst) {
    memcpy(dst, rows + r * kRowS, 64 * sizeof(float));
    dst[64] = blk[kOffC64 + r];
  };
  get_row(R_XPOW, s->xPow);
  get_row(R_DPOW, s->dPow);
  get_row(R_DMINPOW, s->dMinPow);
  get_row(R_DINITMINPOW, s->dInitMinPow);
  get_row(R_SX, s->sx);
  get_row(R_SD, s->sd);
  get_row(R_SE, s->se);
  {
    float tmp[4][65];
    get_row(R_SDE_RE, tmp[0]);
    get_row(R_SDE_IM, tmp[1]);
    get_row(R_SXD_RE, tmp[2]);
    get_row(R_SXD_IM, tmp[3]);
    for (int i = 0; i < 65; ++i) {
      s->sde[i][0] = tmp[0][i];
      s->sde[i][1] = tmp[1][i];
      s->sxd[i][0] = tmp[2][i];
      s->sxd[i][1] = tmp[3][i];
    }
  }
  for (int i = 0; i < kNumPart; ++i) {
    get_row(R_XF_RE + i, s->xfBuf[0] + i * 65);
    get_row(R_XF_IM + i, s->xfBuf[1] + i * 65);
    get_row(R_WF_RE + i, s->wfBuf[0] + i * 65);
    get_row(R_WF_IM + i, s->wfBuf[1] + i * 65);
  }
  s->xfBufBlockPos = b->xf_pos;
  {
    float* raw = &s->xfwBuf[0][0];
    for (int a = 0; a < kNumPartMax; ++a) {
      const int ph = (b->xfw_head + (a == 0 ? 0 : a - 1)) % kNumPartMax;
      get_row(R_XFW + 2 * ph, raw + a * 130);
      get_row(R_XFW + 2 * ph + 1, raw + a * 130 + 65);
    }
  }
  memcpy(s->dBuf, blk + kOffDBuf, 64 * sizeof(float));
  memcpy(s->dBuf + 64, blk + kOffDBuf, 64 * sizeof(float));  // the copy left by aec_core.c:1070
  memcpy(s->eBuf, blk + kOffEBuf, 64 * sizeof(float));
  memcpy(s->eBuf + 64, blk + kOffEBuf, 64 * sizeof(float));
  memcpy(s->outBuf, blk + kOffOutBuf, 64 * sizeof(float));
  memcpy(s->dBufH, blk + kOffDBufH, 64 * sizeof(float));
  memcpy(s->dBufH + 64, blk + kOffDBufH, 64 * sizeof(float));  // the copy left by aec_core.c:1076
  const float* sc = blk + kOffScalars;
  const int32_t* sci = reinterpret_cast<const int32_t*>(sc);
  s->hNlFbMin = sc[S_HNLFBMIN];
  s->hNlFbLocalMin = sc[S_HNLFBLOCALMIN];
  s->hNlXdAvgMin = sc[S_HNLXDAVGMIN];
  s->overDrive = sc[S_OVERDRIVE];
  s->overDriveSm = sc[S_OVERDRIVESM];
  s->hNlNewMin = sci[S_HNLNEWMIN];
  s->hNlMinCtr = sci[S_HNLMINCTR];
  s->delayIdx = sci[S_DELAYIDX];
  s->stNearState = sci[S_STNEARSTATE];
  s->echoState = sci[S_ECHOSTATE];
  s->divergeState = sci[S_DIVERGESTATE];
  s->noiseEstCtr = sci[S_NOISEESTCTR];
  s->delayEstCtr = sci[S_DELAYESTCTR];
  s->seed = reinterpret_cast<const uint32_t*>(sc)[S_SEED];
}

// the far-end pre-buffer and the near / out rings of both bands: device-only data between kOffPre and the rows
// (dBufH, in their middle, belongs to AspAecState)
void keep_rings(float* blk, const float* cur) {
  memcpy(blk + kOffPre, cur + kOffPre, (kOffDBufH - kOffPre) * sizeof(float));
  memcpy(blk + kOffNearFrH, cur + kOffNearFrH, (kOffRows - kOffNearFrH) * sizeof(float));
}

void init_canonical(AspAecState* s) {  // WebRtcAec_InitAec float state, aec_core.c:1562-1610
  memset(s, 0, sizeof *s);
  for (int i = 0; i < 65; i++) s->dMinPow[i] = 1.0e6f;
  for (int i = 0; i < 65; i++) s->sd[i] = 1;
  for (int i = 0; i < 65; i++) s->sx[i] = 1;
  s->hNlFbMin = 1;
  s->hNlFbLocalMin = 1;
  s->hNlXdAvgMin = 1;
  s->overDrive = 2;
  s->overDriveSm = 2;
  s->seed = 777;
}

// WebRtcAec_BufferFarend control plane (echo_cancellation.c:278-339) -> launches on device data.
int flush_pending_farend(AspAecBatch* b) {
  if (b->far_pending) {
    b->far_pending = false;
    if (!b->sim) AEC_TRY(batch_launch_farend(b, b->far_src, b->far_ops));
  }
  return 0;
}

// Delay-agnostic mode with more far-end calls waiting than a descriptor holds: replay them now (no sub-frame follows).
int flush_far_events(AspAecBatch* b) {
  DelayOps d;
  memset(&d, 0, sizeof d);
  d.control = 2;
  d.mult = b->mult;
  d.num_part = b->num_part;
  d.nevents = b->nevents;
  memcpy(d.ev_samples, b->ev_samples, sizeof d.ev_samples);
  memcpy(d.ev_parts, b->ev_parts, sizeof d.ev_parts);
  b->nevents = 0;
  if (!b->sim) AEC_TRY(launch_aec_delay(b->dblocks, b->spectra, b->S, d, b->stream));
  return 0;
}

// defer = true: when the call needs one launch descriptor (the normal case) its device work is
// kept for the following Process launch instead of being launched on its own.
int buffer_farend_device(AspAecBatch* b, const float* far_dev, int n, bool defer = false) {
  {
    const int rc = flush_pending_farend(b);
    if (rc != 0) return rc;
  }
  if (b->skewMode == kAecTrue && b->resample == kAecTrue) {
    // WebRtcAec_ResampleLinear (aec_resampler.c:74-123): the sample positions follow from the skew and the
    // resampler's position alone, so the output length and the new position are computed here, once; the device
    // interpolates every stream's samples with the same float expressions
    const float be = 1 + b->skew;
    int mm = 0;
    float tnew = be * mm + b->rs_position;
    int tn = (int)tnew;
    while (tn < n) {
      mm++;
      tnew = be * mm + b->rs_position;
      tn = (int)tnew;
    }
    if (!b->sim) {
      AEC_TRY(launch_aec_resample(b->rs_buffer, far_dev, b->stage_rs, b->S, n, mm, be, b->rs_position, b->stream));
      far_dev = b->stage_rs;
    }
    b->rs_position += mm * be - n;
    n = mm;
  }
  b->farend_started = 1;
  b->system_delay += n;
  FarOps ops;
  memset(&ops, 0, sizeof ops);
  ops.n = rp_write(&b->pre_pos, n, &ops.wpos);
  bool pending = true;
  int parts_of_call = 0;
  while (rp_avail_read(&b->pre_pos) >= kPartLen2) {
    int rpos;
    rp_read(&b->pre_pos, kPartLen2, &rpos);
    int slot;
    if (b->agn_synced) {
      // the read side (and with it "is the buffer full", aec_core.c:1622-1625) is the stream's own and lives on
      // the device; the write position stays common to the batch: every stream writes one partition here
      slot = b->far_pos.write >= kFarSlots ? b->far_pos.write - kFarSlots : b->far_pos.write;
      b->far_pos.write = slot + 1;
    } else {
      if (rp_avail_write(&b->far_pos) < 1) far_move_read(b, 1);  // aec_core.c:1622-1625
      rp_write(&b->far_pos, 1, &slot);
      if (slot >= kFarSlots) slot -= kFarSlots;
    }
    ++parts_of_call;
    if (ops.nparts == 3) {  // a fourth partition in one call: the full descriptor goes out on its own
      if (!b->sim) AEC_TRY(batch_launch_farend(b, far_dev, ops));
      memset(&ops, 0, sizeof ops);
      pending = false;
    }
    ops.rpos[ops.nparts] = rpos;
    ops.slot[ops.nparts] = slot;
    ops.nparts++;
    rp_move_read(&b->pre_pos, -kPartLen);  // overlap, echo_cancellation.c:336
  }
  if (b->agn_synced) {  // the device replays this call on every stream's own read side at its next control step
    if (b->nevents == kMaxFarEvents) {
      const int rc = flush_far_events(b);
      if (rc != 0) return rc;
    }
    b->ev_samples[b->nevents] = n;
    b->ev_parts[b->nevents] = parts_of_call;
    b->nevents++;
  }
  if (pending || ops.nparts > 0) {
    if (defer && pending) {  // everything of this call fits one descriptor
      b->far_pending = true;
      b->far_ops = ops;
      b->far_src = far_dev;
    } else {
      if (!b->sim) AEC_TRY(batch_launch_farend(b, far_dev, ops));
    }
  }
  return 0;
}

void est_buf_delay_normal(AspAecBatch* b) {  // echo_cancellation.c:816-867
  const int nSampSndCard = b->msInSndCardBuf * kSampMsNb * b->rate_factor;
  int current_delay = nSampSndCard - b->system_delay;
  current_delay += kFrameLen * b->rate_factor;
  if (b->skewMode == kAecTrue && b->resample == kAecTrue) current_delay -= kResamplingDelay;  // echo_cancellation.c:831-833
  if (current_delay < kPartLen) current_delay += far_move_read(b, 1) * kPartLen;
  b->filtDelay = b->filtDelay < 0 ? 0 : b->filtDelay;
  {
    const int16_t v = (int16_t)(0.8 * b->filtDelay + 0.2 * current_delay);
    b->filtDelay = v > 0 ? v : 0;
  }
  const int delay_difference = b->filtDelay - b->knownDelay;
  if (delay_difference > 224) {
    if (b->lastDelayDiff < 96) {
      b->timeForDelayChange = 0;
    } else {
      b->timeForDelayChange++;
    }
  } else if (delay_difference < 96 && b->knownDelay > 0) {
    if (b->lastDelayDiff > 224) {
      b->timeForDelayChange = 0;
    } else {
      b->timeForDelayChange++;
    }
  } else {
    b->timeForDelayChange = 0;
  }
  b->lastDelayDiff = (int16_t)delay_difference;
  if (b->timeForDelayChange > 25) {
    const int v = (int)b->filtDelay - 160;
    b->knownDelay = v > 0 ? v : 0;
  }
}

// WebRtcAec_ProcessFrames control plane (aec_core.c:1647-1778) -> one launch.
// `knownDelay`: the delay ProcessNormal / ProcessExtended hand over (echo_cancellation.c:735-741, 803-812)
int process_frames_device(AspAecBatch* b, const float* near_dev, float* out_dev, int n, int knownDelay) {
  ProcOps ops;
  memset(&ops, 0, sizeof ops);
  const RingPos far_at_entry = b->far_pos;  // what the agnostic mode's first control step starts every stream from
  const int system_delay_at_entry = b->system_delay;
  const int kNumPart = b->num_part;
  ops.mult = b->mult;
  ops.nlp_mode = b->nlp_mode;
  ops.num_high = b->num_high;
  ops.mu = b->extended ? 0.4f : b->normal_mu;                                 // kExtendedMu, aec_core.c:172
  ops.error_threshold = b->extended ? 1.0e-6f : b->normal_error_threshold;    // kExtendedErrorThreshold, :173-175
  for (int j = 0; j < n; j += kFrameLen) {
    SubFrame& sf = ops.sub[ops.nsub++];
    rp_write(&b->near_pos, kFrameLen, &sf.near_wpos);
    if (b->system_delay < kFrameLen) far_move_read(b, -(b->mult + 1));
    if (b->reported_delay_enabled) {  // 2 a) aec_core.c:1703-1718 (2 b, the agnostic mode, runs per stream on the device)
      const int move_elements = (b->core_knownDelay - knownDelay - 32) / kPartLen;
      const int moved_elements = rp_move_read(&b->far_pos, move_elements);
      b->core_knownDelay -= moved_elements * kPartLen;
    }
    while (rp_avail_read(&b->near_pos) >= kPartLen) {
      if (sf.nblocks >= 2) return aec_fail(ASP_ERR_STATE, "more than two blocks in one sub-frame");
      BlockOp& op = sf.blk[sf.nblocks++];
      rp_read(&b->near_pos, kPartLen, &op.near_rpos);
      int slot;
      rp_read(&b->far_pos, 1, &slot);
      op.far_slot = slot >= kFarSlots ? slot - kFarSlots : slot;
      b->xf_pos = b->xf_pos == 0 ? kNumPart - 1 : b->xf_pos - 1;  // aec_core.c:1203-1207
      b->xfw_head = (b->xfw_head + kNumPartMax - 1) % kNumPartMax;
      op.xf_pos = b->xf_pos;
      op.xfw_head = b->xfw_head;
      rp_write(&b->out_pos, kPartLen, &op.out_wpos);
      b->blocks_processed++;
    }
    b->system_delay -= kFrameLen;
    const int out_elements = rp_avail_read(&b->out_pos);
    if (out_elements < kFrameLen) rp_move_read(&b->out_pos, out_elements - kFrameLen);
    rp_read(&b->out_pos, kFrameLen, &sf.out_rpos);
  }
  const float* far_src = nullptr;
  FarOps fops;
  memset(&fops, 0, sizeof fops);
  if (b->far_pending) {
    b->far_pending = false;
    far_src = b->far_src;
    fops = b->far_ops;
  }
  ops.spectra = b->delay_logging;
  if (b->sim) return b->reported_delay_enabled ? 0 : aec_fail(ASP_ERR_STATE, "the delay-agnostic mode needs the device");
  float* met = b->metricsMode ? b->metrics : nullptr;
  if (b->reported_delay_enabled) {
    AEC_TRY(batch_launch_process(b, near_dev, out_dev, n, ops, far_src, fops, b->cur_near_high, b->cur_out_high, met,
                                 b->debug_stamps));
    if (b->delay_logging && !b->flow_rec) {  // the estimator takes the blocks of this call (aec_core.c:1191-1203; hand-off build: aec_flow_flush)
      DelayOps d;
      memset(&d, 0, sizeof d);
      for (int j = 0; j < ops.nsub; ++j) d.npending += ops.sub[j].nblocks;
      d.logging = 1;
      d.mult = b->mult;
      d.num_part = b->num_part;
      if (d.npending > 0) AEC_TRY(launch_aec_delay(b->dblocks, b->spectra, b->S, d, b->stream));
    }
    return 0;
  }
  // ---- delay-agnostic mode: per 80-sample sub-frame the per-stream control step (with the estimator's share of the
  // previous sub-frame's blocks), then the blocks themselves with the far slots that step chose
  ops.agnostic = 1;
  if (agn_fused(b)) {
    // one launch per call: every stream's wave runs its own control steps and the estimator between its blocks
    // (aec_kernels.hip, process_call<AGN>)
    AgnOps agn;
    memset(&agn, 0, sizeof agn);
    agn.logging = b->delay_logging;
    for (int j = 0; j < ops.nsub; ++j) {
      DelayOps& d = agn.sub[j];
      d.control = 1;
      d.mult = b->mult;
      d.num_part = b->num_part;
      d.nblocks = ops.sub[j].nblocks;
      if (!b->agn_synced) {  // the first sub-frame after Init: every stream starts from the batch's values
        d.sync = 1;
        d.h_far_read = far_at_entry.read;
        d.h_far_write = far_at_entry.write;
        d.h_far_wrap = far_at_entry.wrap;
        d.h_system_delay = system_delay_at_entry;
        b->agn_synced = true;
        b->nevents = 0;
      } else if (j == 0) {
        d.nevents = b->nevents;
        memcpy(d.ev_samples, b->ev_samples, sizeof d.ev_samples);
        memcpy(d.ev_parts, b->ev_parts, sizeof d.ev_parts);
        b->nevents = 0;
      }
    }
    if (b->flow_rec) return aec_flow_record(b, near_dev, out_dev, n, ops, far_src, fops, &agn);
    if (!b->dual)
      AEC_TRY(launch_aec_process_agn(b->state, b->far_ring, b->tables, near_dev, out_dev, b->S, n, ops, far_src, fops,
                                     b->dblocks, agn, b->stream, 0, -1, b->num_part));
    else {
      const int half = ((b->S / 2 + 3) / 4) * 4;
      AEC_TRY(launch_aec_process_agn(b->state, b->far_ring, b->tables, near_dev, out_dev, b->S, n, ops, far_src, fops,
                                     b->dblocks, agn, b->stream, 0, half, b->num_part));
      AEC_TRY(launch_aec_process_agn(b->state, b->far_ring, b->tables, near_dev, out_dev, b->S, n, ops, far_src, fops,
                                     b->dblocks, agn, b->side, half, b->S, b->num_part));
    }
    return 0;
  }
  int prev_blocks = 0;
  for (int j = 0; j < ops.nsub; ++j) {
    DelayOps d;
    memset(&d, 0, sizeof d);
    d.npending = b->delay_logging ? prev_blocks : 0;
    d.logging = b->delay_logging;
    d.control = 1;
    d.mult = b->mult;
    d.num_part = b->num_part;
    d.nblocks = ops.sub[j].nblocks;
    if (!b->agn_synced) {  // the first sub-frame after Init: every stream starts from the batch's values
      d.sync = 1;
      d.h_far_read = far_at_entry.read;
      d.h_far_write = far_at_entry.write;
      d.h_far_wrap = far_at_entry.wrap;
      d.h_system_delay = system_delay_at_entry;
      b->agn_synced = true;
      b->nevents = 0;
    } else if (j == 0) {
      d.nevents = b->nevents;
      memcpy(d.ev_samples, b->ev_samples, sizeof d.ev_samples);
      memcpy(d.ev_parts, b->ev_parts, sizeof d.ev_parts);
      b->nevents = 0;
    }
    AEC_TRY(launch_aec_delay(b->dblocks, b->spectra, b->S, d, b->stream));
    ProcOps one = ops;
    one.nsub = 1;
    one.sub[0] = ops.sub[j];
    const size_t off = (size_t)kFrameLen * j;
    AEC_TRY(batch_launch_process(b, near_dev + off, out_dev + off, n, one, j == 0 ? far_src : nullptr, fops,
                                 b->cur_near_high ? b->cur_near_high + off : nullptr,
                                 b->cur_out_high ? b->cur_out_high + off : nullptr, met, j == 0 ? b->debug_stamps : nullptr));
    prev_blocks = ops.sub[j].nblocks;
  }
  if (b->delay_logging && prev_blocks > 0) {  // the last sub-frame's blocks
    DelayOps d;
    memset(&d, 0, sizeof d);
    d.npending = prev_blocks;
    d.logging = 1;
    d.mult = b->mult;
    d.num_part = b->num_part;
    AEC_TRY(launch_aec_delay(b->dblocks, b->spectra, b->S, d, b->stream));
  }
  return 0;
}

// ProcessNormal (echo_cancellation.c:594-742) on device buffers.
int process_normal_device(AspAecBatch* b, const float* near_dev, float* out_dev, int n,
                          int16_t msInSndCardBuf, int32_t skew, int* rc_ref) {
  const int nBlocks10ms = n / (kFrameLen * b->rate_factor);
  msInSndCardBuf = msInSndCardBuf > kMaxTrustedDelayMs ? kMaxTrustedDelayMs : msInSndCardBuf;
  msInSndCardBuf += 10;
  b->msInSndCardBuf = msInSndCardBuf;
  if (b->skewMode == kAecTrue) {  // echo_cancellation.c:614-645
    if (b->skewFrCtr < 25) {
      b->skewFrCtr++;
    } else {
      int err = 0;  // WebRtcAec_GetSkew, aec_resampler.c:125-141
      if (b->rs_skewDataIndex < kSkewEstimateFrames) {
        b->rs_skewData[b->rs_skewDataIndex] = skew;
        b->rs_skewDataIndex++;
      } else if (b->rs_skewDataIndex == kSkewEstimateFrames) {
        err = estimate_skew(b->rs_skewData, kSkewEstimateFrames, b->scSampFreq, &b->skew);
        b->rs_skewEstimate = b->skew;
        b->rs_skewDataIndex++;
      } else {
        b->skew = b->rs_skewEstimate;
      }
      if (err == -1) {
        b->skew = 0;
        b->lastError = AEC_BAD_PARAMETER_WARNING;
        *rc_ref = -1;
      }
      b->skew /= b->sampFactor * n;
      if (b->skew < 1.0e-3 && b->skew > -1.0e-3) {
        b->resample = kAecFalse;
      } else {
        b->resample = kAecTrue;
      }
      if (b->skew < -0.5f) {
        b->skew = -0.5f;
      } else if (b->skew > 1.0f) {
        b->skew = 1.0f;
      }
    }
  }
  if (b->startup_phase) {
    {
      int rc = flush_pending_farend(b);
      if (rc == 0 && b->flow_rec) rc = aec_flow_flush(b);  // the pass-through copies below follow the recorded steps
      if (rc != 0) return rc;
    }
    if (b->vrec) {
      b->vproc_host[b->vstream].mode = 0;  // this stream passes its near end through (the kernel copies it)
    } else {
      if (near_dev != out_dev && !b->sim)
        AEC_TRY(hipMemcpyAsync(out_dev, near_dev, (size_t)b->S * n * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
      if (b->num_high > 0 && b->cur_near_high != b->cur_out_high && !b->sim)
        AEC_TRY(hipMemcpyAsync(b->cur_out_high, b->cur_near_high, (size_t)b->S * n * sizeof(float),
                               hipMemcpyDeviceToDevice, b->stream));
    }
    if (b->checkBuffSize) {
      b->checkBufSizeCtr++;
      if (b->counter == 0) {
        b->firstVal = b->msInSndCardBuf;
        b->sum = 0;
      }
      double lim = 0.2 * b->msInSndCardBuf;
      if (lim < kSampMsNb) lim = kSampMsNb;
      if (abs(b->firstVal - b->msInSndCardBuf) < lim) {
        b->sum += b->msInSndCardBuf;
        b->counter++;
      } else {
        b->counter = 0;
      }
      if (b->counter * nBlocks10ms >= 6) {
        const int v = (3 * b->sum * b->rate_factor * 8) / (4 * b->counter * kPartLen);
        b->bufSizeStart = v < kMaxBufSizeStart ? v : kMaxBufSizeStart;
        b->checkBuffSize = 0;
      }
      if (b->checkBufSizeCtr * nBlocks10ms > 50) {
        const int v = (b->msInSndCardBuf * b->rate_factor * 3) / 40;
        b->bufSizeStart = v < kMaxBufSizeStart ? v : kMaxBufSizeStart;
        b->checkBuffSize = 0;
      }
    }
    if (!b->checkBuffSize) {
      const int overhead_elements = b->system_delay / kPartLen - b->bufSizeStart;
      if (overhead_elements == 0) {
        b->startup_phase = 0;
      } else if (overhead_elements > 0) {
        far_move_read(b, overhead_elements);
        b->startup_phase = 0;
      }
    }
    return 0;
  }
  if (b->reported_delay_enabled) est_buf_delay_normal(b);  // echo_cancellation.c:725-727
  return process_frames_device(b, near_dev, out_dev, n, b->knownDelay);
}

void est_buf_delay_extended(AspAecBatch* b) {  // EstBufDelayExtended, echo_cancellation.c:869-922
  const int reported_delay = b->msInSndCardBuf * kSampMsNb * b->rate_factor;
  int current_delay = reported_delay - b->system_delay;
  current_delay += kFrameLen * b->rate_factor;
  if (b->skewMode == kAecTrue && b->resample == kAecTrue) current_delay -= kResamplingDelay;  // :884-886
  if (current_delay < kPartLen) current_delay += far_move_read(b, 2) * kPartLen;
  if (b->filtDelay == -1) {
    const double v = 0.5 * current_delay;
    b->filtDelay = (int16_t)(v > 0 ? v : 0);
  } else {
    const int16_t v = (int16_t)(0.95 * b->filtDelay + 0.05 * current_delay);
    b->filtDelay = v > 0 ? v : 0;
  }
  const int delay_difference = b->filtDelay - b->knownDelay;
  if (delay_difference > 384) {
    if (b->lastDelayDiff < 128) {
      b->timeForDelayChange = 0;
    } else {
      b->timeForDelayChange++;
    }
  } else if (delay_difference < 128 && b->knownDelay > 0) {
    if (b->lastDelayDiff > 384) {
      b->timeForDelayChange = 0;
    } else {
      b->timeForDelayChange++;
    }
  } else {
    b->timeForDelayChange = 0;
  }
  b->lastDelayDiff = (int16_t)delay_difference;
  if (b->timeForDelayChange > 25) {
    const int v = (int)b->filtDelay - 256;
    b->knownDelay = v > 0 ? v : 0;
  }
}

// ProcessExtended (echo_cancellation.c:744-814) on device buffers: the trusted-delay build (neither
// WEBRTC_UNTRUSTED_DELAY nor WEBRTC_MAC: kFixedDelayMs 50, kMinTrustedDelayMs 20, kDelayDiffOffsetSamples 0, :70-84).
int process_extended_device(AspAecBatch* b, const float* near_dev, float* out_dev, int n, int16_t reported_delay_ms) {
  const int kFixedDelayMs = 50, kMinTrustedDelayMs = 20, kDelayDiffOffsetSamples = 0;
  reported_delay_ms = reported_delay_ms < kMinTrustedDelayMs ? kMinTrustedDelayMs : reported_delay_ms;
  reported_delay_ms = reported_delay_ms >= kMaxTrustedDelayMs ? kFixedDelayMs : reported_delay_ms;
  b->msInSndCardBuf = reported_delay_ms;
  if (!b->farend_started) {  // pass the near end through until the far end starts (:768-776)
    if (b->flow_rec) {
      const int rc = aec_flow_flush(b);
      if (rc != 0) return rc;
    }
    if (b->vrec) {
      b->vproc_host[b->vstream].mode = 0;
      return 0;
    }
    if (near_dev != out_dev && !b->sim)
      AEC_TRY(hipMemcpyAsync(out_dev, near_dev, (size_t)b->S * n * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
    if (b->num_high > 0 && b->cur_near_high != b->cur_out_high && !b->sim)
      AEC_TRY(hipMemcpyAsync(b->cur_out_high, b->cur_near_high, (size_t)b->S * n * sizeof(float),
                             hipMemcpyDeviceToDevice, b->stream));
    return 0;
  }
  if (b->startup_phase) {  // :777-794
    const int startup_size_ms = reported_delay_ms < kFixedDelayMs ? kFixedDelayMs : reported_delay_ms;
    const int overhead_elements = (b->system_delay - startup_size_ms / 2 * b->rate_factor * 8) / kPartLen;
    far_move_read(b, overhead_elements);
    b->startup_phase = 0;
  }
  if (b->reported_delay_enabled) est_buf_delay_extended(b);  // echo_cancellation.c:796-798
  const int adjusted = b->knownDelay + kDelayDiffOffsetSamples;
  return process_frames_device(b, near_dev, out_dev, n, adjusted > 0 ? adjusted : 0);
}

// WebRtcAec_Process checks (echo_cancellation.c:341-375); *rc is the reference's return value.
int process_device(AspAecBatch* b, const float* near_dev, float* out_dev, int n, int msInSndCardBuf,
                   int* rc, int32_t skew = 0) {
  *rc = 0;
  if (msInSndCardBuf < 0) {
    msInSndCardBuf = 0;
    b->lastError = AEC_BAD_PARAMETER_WARNING;
    *rc = -1;
  } else if (msInSndCardBuf > kMaxTrustedDelayMs) {
    b->lastError = AEC_BAD_PARAMETER_WARNING;
    *rc = -1;
  }
  if (b->extended) return process_extended_device(b, near_dev, out_dev, n, (int16_t)msInSndCardBuf);  // :377-394
  return process_normal_device(b, near_dev, out_dev, n, (int16_t)msInSndCardBuf, skew, rc);
}

int check_running(AspAecBatch* b, const void* p, int n) {
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (p == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (b->initFlag != kInitCheck) {
    b->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (n != 80 && n != 160) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  return 0;
}

// ------------------------------------------------------------------ per-stream control
// The reference takes the reported delay, Init and the call pattern per handle (echo_cancellation.c:341-347, 196-276).
// Once a caller uses AspAecBatch_ProcessV / _InitStream the batch keeps one AecCtl per stream (`per`); every call
// then runs the control functions above once per stream -- on that stream's copy, swapped into the handle's base --
// with the launches they would issue recorded as that stream's descriptor; one far-end launch and one Process launch
// per call read the descriptors from device memory (aec_farend_v_kernel, aec_process_v_kernel).
bool vmode(const AspAecBatch* b) { return !b->per.empty(); }
void ctl_load(AspAecBatch* b, int s) { static_cast<AecCtl&>(*b) = b->per[(size_t)s]; }
void ctl_store(AspAecBatch* b, int s) { b->per[(size_t)s] = static_cast<const AecCtl&>(*b); }

// one stream's control plane in the handle's base for the length of a scope (ExportState / ImportState / GetControlStream)
struct StreamCtlScope {
  AspAecBatch* b;
  int s;
  AecCtl saved;
  StreamCtlScope(AspAecBatch* b_, int s_) : b(b_), s(s_) {
    if (vmode(b)) {
      saved = static_cast<const AecCtl&>(*b);
      ctl_load(b, s);
    }
  }
  ~StreamCtlScope() {
    if (vmode(b)) {
      ctl_store(b, s);
      static_cast<AecCtl&>(*b) = saved;
    }
  }
};

int enter_vmode(AspAecBatch* b) {
  if (vmode(b)) return 0;
  if (b->sim) return aec_fail(ASP_ERR_STATE, "per-stream control: control-only handle");
  if (b->num_high > 0 || b->delay_logging || !b->reported_delay_enabled || b->skewMode == kAecTrue || b->metricsMode)
    return aec_fail(ASP_ERR_STATE, "per-stream control covers the one-band configuration with reported delays "
                                   "(no delay logging / delay-agnostic mode / skew compensation / metrics)");
  {
    const int rc = flush_pending_farend(b);
    if (rc != 0) return rc;
  }
  if (!b->vproc_host) {
    AEC_TRY(hipHostMalloc((void**)&b->vproc_host, sizeof(AecStreamStep) * (size_t)b->S, hipHostMallocDefault));
    AEC_TRY(hipMalloc((void**)&b->vproc_dev, sizeof(AecStreamStep) * (size_t)b->S));
    AEC_TRY(hipHostMalloc((void**)&b->vfar_host, sizeof(FarOps) * (size_t)b->S, hipHostMallocDefault));
    AEC_TRY(hipMalloc((void**)&b->vfar_dev, sizeof(FarOps) * (size_t)b->S));
    AEC_TRY(hipEventCreateWithFlags(&b->vev, hipEventDisableTiming));
  }
  b->per.assign((size_t)b->S, static_cast<const AecCtl&>(*b));
  return 0;
}

// WebRtcAec_BufferFarend of every stream, each on its own control plane
int buffer_farend_v(AspAecBatch* b, const float* far_dev, int n) {
  AEC_TRY(hipEventSynchronize(b->vev));  // the previous launches have read their descriptors
  int err = 0;
  for (int s = 0; s < b->S && err == 0; ++s) {
    ctl_load(b, s);
    memset(&b->vfar_host[s], 0, sizeof(FarOps));
    b->vrec = true;
    b->vstream = s;
    b->vfar_recorded = 0;
    err = buffer_farend_device(b, far_dev, n, false);
    b->vrec = false;
    ctl_store(b, s);
  }
  ctl_load(b, 0);
  if (err != 0) return err;
  AEC_TRY(hipMemcpyAsync(b->vfar_dev, b->vfar_host, sizeof(FarOps) * (size_t)b->S, hipMemcpyHostToDevice, b->stream));
  AEC_TRY(launch_aec_farend_v(b->state, b->far_ring, b->tables, far_dev, b->S, b->vfar_dev, n, b->num_part, b->stream));
  AEC_TRY(hipEventRecord(b->vev, b->stream));
  return 0;
}

// WebRtcAec_Process of every stream with its own reported delay; status[s] (may be null) receives the reference's
// return value of stream s (0, or -1 with lastError set: a delay outside 0 .. 500 ms still processes)
int process_v(AspAecBatch* b, const float* near_dev, float* out_dev, int n, const int16_t* ms, int ms_stride,
              int32_t* status, int* any_rc) {
  AEC_TRY(hipEventSynchronize(b->vev));
  int err = 0;
  *any_rc = 0;
  for (int s = 0; s < b->S && err == 0; ++s) {
    ctl_load(b, s);
    b->vproc_host[s].mode = 0;
    b->vrec = true;
    b->vstream = s;
    int rc = 0;
    err = process_device(b, near_dev, out_dev, n, ms[(size_t)s * ms_stride], &rc, 0);
    b->vrec = false;
    ctl_store(b, s);
    if (status) status[s] = rc;
    *any_rc |= rc;
  }
  ctl_load(b, 0);
  if (err != 0) return err;
  AEC_TRY(hipMemcpyAsync(b->vproc_dev, b->vproc_host, sizeof(AecStreamStep) * (size_t)b->S, hipMemcpyHostToDevice, b->stream));
  AEC_TRY(launch_aec_process_v(b->state, b->far_ring, b->tables, near_dev, out_dev, b->S, n, b->vproc_dev, b->num_part,
                               b->stream));
  AEC_TRY(hipEventRecord(b->vev, b->stream));
  return 0;
}

// BufferFarend + Process of one frame on device buffers (Run / TimedSteps): the batch's one control plane with the
// far-end work deferred into the Process launch, or every stream's own
int frame_device(AspAecBatch* b, const float* far_dev, const float* near_dev, float* out_dev, int n, int msInSndCardBuf, int* rc) {
  if (vmode(b)) {
    int err = buffer_farend_v(b, far_dev, n);
    if (err != 0) return err;
    const int16_t ms16 = (int16_t)(msInSndCardBuf < -32768 ? -32768 : msInSndCardBuf > 32767 ? 32767 : msInSndCardBuf);
    return process_v(b, near_dev, out_dev, n, &ms16, 0, nullptr, rc);
  }
  const int err = buffer_farend_device(b, far_dev, n, true);
  if (err != 0) return err;
  return process_device(b, near_dev, out_dev, n, msInSndCardBuf, rc);
}

}  // namespace

extern "C" {

int AspAecBatch_Create(AspAecBatch** out, int num_streams, int device) {
  AspDeviceScope dev_scope_;
  if (!out || num_streams <= 0) return aec_fail(ASP_ERR_PARAM, "AspAecBatch_Create: bad argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return aec_fail(ASP_ERR_NO_DEVICE, "no HIP device: the echo canceller has no CPU fallback");
  if (device < 0 || device >= count) return aec_fail(ASP_ERR_PARAM, "device ordinal out of range");
  AEC_TRY(hipSetDevice(device));
  AspAecBatch* b = new AspAecBatch();
  b->S = num_streams;
  b->device = device;
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  b->own_stream = e == hipSuccess;
  if (e == hipSuccess) e = hipMalloc((void**)&b->state, state_bytes(b));
  if (e == hipSuccess) e = hipMalloc((void**)&b->far_ring, far_bytes(b));
  if (e == hipSuccess) e = hipMalloc((void**)&b->tables, sizeof(AecTables));
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_far, (size_t)num_streams * 160 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_near, (size_t)num_streams * 160 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_out, (size_t)num_streams * 160 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_near_h, (size_t)num_streams * 160 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_out_h, (size_t)num_streams * 160 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->metrics, (size_t)num_streams * kMetDwords * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->rs_buffer, (size_t)num_streams * kResamplerBufferSize * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->stage_rs, (size_t)num_streams * kResamplerBufferSize * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->dblocks, (size_t)num_streams * sizeof(DelayBlock));
  if (e == hipSuccess) e = hipMalloc((void**)&b->spectra, (size_t)num_streams * kSpecBlocks * kSpecDwords * sizeof(float));
  if (e == hipSuccess) {  // WebRtc_set_lookahead at Create (aec_core.c:1374-1378); the Init functions leave it alone
    std::vector<DelayBlock> all((size_t)num_streams);
    memset(all.data(), 0, all.size() * sizeof(DelayBlock));
    for (auto& d : all) d.s.lookahead = ASP_AEC_DELAY_LOOKAHEAD;
    e = hipMemcpy(b->dblocks, all.data(), all.size() * sizeof(DelayBlock), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipEventCreate(&b->ev0);
  if (e == hipSuccess) e = hipEventCreate(&b->ev1);
  if (e == hipSuccess) {
    AecTables T;
    build_tables(&T);
    e = hipMemcpy(b->tables, &T, sizeof T, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    AspAecBatch_Free(b);
    return aec_fail(ASP_ERR_HIP, "AspAecBatch_Create", e);
  }
  *out = b;
  return ASP_OK;
}

// The integer control plane alone (no device is touched, nothing is launched): BufferFarend /
// Process / set_config / GetControl behave as on a real batch, so the host logic can be checked
// against the oracle on a machine without a GPU.  Every data-touching entry point refuses it.
int AspAecBatch_CreateControlOnly(AspAecBatch** out, int num_streams) {
  AspDeviceScope dev_scope_;
  if (!out || num_streams <= 0) return aec_fail(ASP_ERR_PARAM, "AspAecBatch_CreateControlOnly: bad argument");
  AspAecBatch* b = new AspAecBatch();
  b->S = num_streams;
  b->sim = true;
  *out = b;
  return ASP_OK;
}

int AspAecBatch_Free(AspAecBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return -1;
  if (b->sim) {
    delete b;
    return 0;
  }
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->state) (void)hipFree(b->state);
  if (b->far_ring) (void)hipFree(b->far_ring);
  if (b->tables) (void)hipFree(b->tables);
  if (b->stage_far) (void)hipFree(b->stage_far);
  if (b->stage_near) (void)hipFree(b->stage_near);
  if (b->stage_out) (void)hipFree(b->stage_out);
  if (b->stage_near_h) (void)hipFree(b->stage_near_h);
  if (b->stage_out_h) (void)hipFree(b->stage_out_h);
  if (b->metrics) (void)hipFree(b->metrics);
  if (b->dblocks) (void)hipFree(b->dblocks);
  if (b->rs_buffer) (void)hipFree(b->rs_buffer);
  if (b->stage_rs) (void)hipFree(b->stage_rs);
  if (b->spectra) (void)hipFree(b->spectra);
  if (b->vproc_host) (void)hipHostFree(b->vproc_host);
  if (b->vproc_dev) (void)hipFree(b->vproc_dev);
  if (b->vfar_host) (void)hipHostFree(b->vfar_host);
  if (b->vfar_dev) (void)hipFree(b->vfar_dev);
  if (b->vev) (void)hipEventDestroy(b->vev);
  if (b->flow_seq) (void)hipFree(b->flow_seq);
  if (b->flow_bits) (void)hipFree(b->flow_bits);
  if (b->flow_abort) (void)hipFree(b->flow_abort);
  if (b->flow_dev) (void)hipFree(b->flow_dev);
  if (b->flow_host) (void)hipHostFree(b->flow_host);
  for (int i = 0; i < 4; ++i)
    if (b->flow_ev[i]) (void)hipEventDestroy(b->flow_ev[i]);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->side) {
    (void)hipStreamSynchronize(b->side);
    (void)hipStreamDestroy(b->side);
    (void)hipEventDestroy(b->ev_fork);
    (void)hipEventDestroy(b->ev_join);
  }
  if (b->stream && b->own_stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return 0;
}

int AspAecBatch_num_streams(const AspAecBatch* b) { return b ? b->S : 0; }
int AspAecBatch_get_error_code(const AspAecBatch* b) { return b ? b->lastError : AEC_NULL_POINTER_ERROR; }

int AspAecBatch_set_config(AspAecBatch* b, AecConfig config) {  // echo_cancellation.c:410-438
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (b->initFlag != kInitCheck) {
    b->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (config.skewMode != kAecFalse && config.skewMode != kAecTrue) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  b->skewMode = config.skewMode;
  if (config.nlpMode != kAecNlpConservative && config.nlpMode != kAecNlpModerate &&
      config.nlpMode != kAecNlpAggressive) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (config.metricsMode != kAecFalse && config.metricsMode != kAecTrue) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (config.delay_logging != kAecFalse && config.delay_logging != kAecTrue) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (!b->per.empty()) {  // per-stream control: the suppression level of every stream; the optional modes are not covered
    if (config.skewMode == kAecTrue || config.metricsMode == kAecTrue || config.delay_logging == kAecTrue)
      return aec_fail(ASP_ERR_STATE, "set_config: per-stream control covers nlpMode only; Init the batch to switch modes");
    for (auto& c : b->per) c.nlp_mode = config.nlpMode;
  }
  b->nlp_mode = config.nlpMode;  // WebRtcAec_SetConfigCore, aec_core.c:1844-1862
  b->metricsMode = config.metricsMode;
  if (b->metricsMode && !b->sim) {
    const int err = init_metrics_device(b);
    if (err) return err;
  }
  b->delay_logging = config.delay_logging;
  if (b->delay_logging && !b->sim) {  // memset(delay_histogram), aec_core.c:1858-1860
    AEC_TRY(hipSetDevice(b->device));
    AEC_TRY(hipMemset2DAsync(reinterpret_cast<char*>(b->dblocks) + offsetof(DelayBlock, s.delay_histogram), sizeof(DelayBlock), 0,
                             sizeof(((AspAecDelayState*)nullptr)->delay_histogram), (size_t)b->S, b->stream));
  }
  return 0;
}

// The control plane as WebRtcAec_Init leaves it (echo_cancellation.c:196-276; WebRtcAec_InitAec, aec_core.c:1460-1615;
// the default configuration of :261-270): host integers only.
static void init_control(AspAecBatch* b, int32_t sampFreq, int32_t scSampFreq) {
  b->sampFreq = sampFreq;
  b->scSampFreq = scSampFreq;
  if (sampFreq == 8000) {
    b->normal_mu = 0.6f;
    b->normal_error_threshold = 2e-6f;
  } else {
    b->normal_mu = 0.5f;
    b->normal_error_threshold = 1.5e-6f;
  }
  rp_init(&b->near_pos, kFrBufLen);
  rp_init(&b->out_pos, kFrBufLen);
  rp_init(&b->far_pos, kFarSlots);
  b->system_delay = 0;
  b->num_high = sampFreq == 32000 ? 1 : 0;             // aec_core.c:1466-1473
  b->mult = sampFreq == 32000 ? 2 : sampFreq / 8000;    // aec_core.c:1541-1545
  b->core_knownDelay = 0;
  b->xf_pos = 0;
  b->xfw_head = 0;
  b->blocks_processed = 0;
  b->extended = 0;  // aec_core.c:1522-1523
  b->num_part = kNumPartNormal;
  b->rs_position = 0.f;
  b->rs_skewDataIndex = 0;
  b->rs_skewEstimate = 0.f;
  b->skewFrCtr = 0;  // echo_cancellation.c:256-259
  b->resample = kAecFalse;
  b->skew = 0.f;
  b->reported_delay_enabled = 1;  // aec_core.c:1517-1521 (not Android)
  b->agn_synced = false;
  b->nevents = 0;
  rp_init(&b->pre_pos, kPreLen);
  rp_move_read(&b->pre_pos, -kPartLen);  // start overlap, echo_cancellation.c:226
  b->initFlag = kInitCheck;
  b->splitSampFreq = sampFreq == 32000 ? 16000 : sampFreq;  // echo_cancellation.c:231-235
  b->rate_factor = b->splitSampFreq / 8000;
  b->sampFactor = (b->scSampFreq * 1.0f) / b->splitSampFreq;  // echo_cancellation.c:237
  b->sum = 0;
  b->counter = 0;
  b->checkBuffSize = 1;
  b->firstVal = 0;
  b->startup_phase = 1;  // = reported_delay_enabled, which InitAec has just set (echo_cancellation.c:247)
  b->bufSizeStart = 0;
  b->checkBufSizeCtr = 0;
  b->msInSndCardBuf = 0;
  b->filtDelay = -1;
  b->timeForDelayChange = 0;
  b->knownDelay = 0;
  b->lastDelayDiff = 0;
  b->farend_started = 0;
  b->far_pending = false;
  // the default configuration (echo_cancellation.c:261-270 through WebRtcAec_set_config)
  b->nlp_mode = kAecNlpModerate;
  b->skewMode = kAecFalse;
  b->metricsMode = kAecFalse;
  b->delay_logging = kAecFalse;
}

int AspAecBatch_Init(AspAecBatch* b, int32_t sampFreq, int32_t scSampFreq) {  // echo_cancellation.c:196-276
  AspDeviceScope dev_scope_;
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (sampFreq != 8000 && sampFreq != 16000 && sampFreq != 32000 && sampFreq != 48000) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (sampFreq > 32000) {  // 48 kHz: the reference's own mult breaks there (aec_core.c:1541-1543)
    b->lastError = AEC_UNSUPPORTED_FUNCTION_ERROR;
    return -1;
  }
  b->sampFreq = sampFreq;
  if (scSampFreq < 1 || scSampFreq > 96000) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  b->per.clear();  // back to one control plane for the batch
  const int old_num_part = b->num_part;
  init_control(b, sampFreq, scSampFreq);
  memset(b->rs_skewData, 0, sizeof b->rs_skewData);
  if (!b->sim) {
    AEC_TRY(hipSetDevice(b->device));
    AEC_TRY(hipStreamSynchronize(b->stream));
    if (old_num_part != kNumPartNormal) {  // back to the 12-partition blocks
      AEC_TRY(hipFree(b->state));
      b->state = nullptr;
      AEC_TRY(hipMalloc((void**)&b->state, state_bytes(b)));
    }
    const int kStateDwords = state_dwords(b);
    std::vector<AspAecState> s0(1);  // ~50 KB, per call: two batches may be initialised from two host threads at once
    init_canonical(s0.data());
    std::vector<float> blk(kStateDwords);
    pack_stream(b, s0.data(), blk.data());
    std::vector<float> all((size_t)b->S * kStateDwords);
    for (int s = 0; s < b->S; ++s) memcpy(all.data() + (size_t)s * kStateDwords, blk.data(), kStateDwords * sizeof(float));
    AEC_TRY(hipMemcpy(b->state, all.data(), state_bytes(b), hipMemcpyHostToDevice));
    // on the batch's own (non-blocking) stream: a null-stream memset is not ordered with its kernels
    AEC_TRY(hipMemsetAsync(b->far_ring, 0, far_bytes(b), b->stream));  // WebRtc_InitBuffer zeroes the rings
    AEC_TRY(hipStreamSynchronize(b->stream));
    const int err = init_metrics_device(b);  // aec_core.c:1612-1613
    if (err) return err;
    const int err2 = init_delay_device(b);   // aec_core.c:1502-1516
    if (err2) return err2;
    // WebRtcAec_InitResampler (echo_cancellation.c:221, aec_resampler.c:55-66)
    AEC_TRY(hipMemsetAsync(b->rs_buffer, 0, (size_t)b->S * kResamplerBufferSize * sizeof(float), b->stream));
  }
  return 0;
}

int AspAecBatch_BufferFarend(AspAecBatch* b, const float* farend, int nrOfSamples, int mem) {
  AspDeviceScope dev_scope_;
  const int chk = check_running(b, farend, nrOfSamples);
  if (chk != 0) return chk;
  if (b->sim) return buffer_farend_device(b, farend, nrOfSamples);
  AEC_TRY(hipSetDevice(b->device));
  const float* dev = farend;
  if (mem == ASP_MEM_HOST) {
    AEC_TRY(hipMemcpyAsync(b->stage_far, farend, (size_t)b->S * nrOfSamples * sizeof(float), hipMemcpyHostToDevice, b->stream));
    dev = b->stage_far;
  }
  const int rc = vmode(b) ? buffer_farend_v(b, dev, nrOfSamples) : buffer_farend_device(b, dev, nrOfSamples);
  if (rc != 0) return rc;
  if (mem == ASP_MEM_HOST) AEC_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

static int process_impl(AspAecBatch* b, const float* nearend, const float* near_high, float* out,
                        float* out_high, int nrOfSamples, int msInSndCardBuf, int mem, int32_t skew);

int AspAecBatch_Process(AspAecBatch* b, const float* nearend, float* out, int nrOfSamples,
                        int msInSndCardBuf, int32_t skew, int mem) {
  AspDeviceScope dev_scope_;
  if (b && b->num_high > 0) {
    b->lastError = AEC_BAD_PARAMETER_ERROR;  // a 32 kHz batch needs both bands (ProcessBands)
    return -1;
  }
  return process_impl(b, nearend, nullptr, out, nullptr, nrOfSamples, msInSndCardBuf, mem, skew);
}

int AspAecBatch_ProcessBands(AspAecBatch* b, const float* near_low, const float* near_high,
                             float* out_low, float* out_high, int nrOfSamples, int msInSndCardBuf,
                             int32_t skew, int mem) {
  AspDeviceScope dev_scope_;
  if (b && (b->num_high < 1 || near_high == nullptr || out_high == nullptr)) {
    b->lastError = b->num_high < 1 ? AEC_BAD_PARAMETER_ERROR : AEC_NULL_POINTER_ERROR;
    return -1;
  }
  return process_impl(b, near_low, near_high, out_low, out_high, nrOfSamples, msInSndCardBuf, mem, skew);
}

int AspAecBatch_num_bands(const AspAecBatch* b) { return b ? 1 + b->num_high : 0; }

// WebRtcAec_Process of every stream with the stream's OWN reported delay (echo_cancellation.c:341-347 takes it per
// handle): msInSndCardBuf[num_streams]; skew[num_streams] may be null (skew compensation is not covered by the
// per-stream control); status[num_streams] (may be null) receives each stream's reference return value.
int AspAecBatch_ProcessV(AspAecBatch* b, const float* nearend, float* out, int nrOfSamples, const int16_t* msInSndCardBuf,
                         const int32_t* skew, int32_t* status, int mem) {
  AspDeviceScope dev_scope_;
  (void)skew;
  if (b && out == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  const int chk = check_running(b, nearend, nrOfSamples);
  if (chk != 0) return chk;
  if (!msInSndCardBuf) return aec_fail(ASP_ERR_PARAM, "ProcessV: null delay array");
  if (mem != ASP_MEM_HOST && mem != ASP_MEM_DEVICE) return aec_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  AEC_TRY(hipSetDevice(b->device));
  {
    const int rc = enter_vmode(b);
    if (rc != 0) return rc;
  }
  const size_t bytes = (size_t)b->S * nrOfSamples * sizeof(float);
  const float* nd = nearend;
  float* od = out;
  if (mem == ASP_MEM_HOST) {
    AEC_TRY(hipMemcpyAsync(b->stage_near, nearend, bytes, hipMemcpyHostToDevice, b->stream));
    nd = b->stage_near;
    od = b->stage_out;
  }
  int rc = 0;
  const int err = process_v(b, nd, od, nrOfSamples, msInSndCardBuf, 1, status, &rc);
  if (err != 0) return err;
  if (mem == ASP_MEM_HOST) {
    AEC_TRY(hipMemcpyAsync(out, od, bytes, hipMemcpyDeviceToHost, b->stream));
    AEC_TRY(hipStreamSynchronize(b->stream));
  }
  return rc;
}

static void init_control(AspAecBatch* b, int32_t sampFreq, int32_t scSampFreq);

// WebRtcAec_Init of ONE stream of a running batch (echo_cancellation.c:196-276 per handle): its control plane, its
// state block and its far-ring slots go back to their initial values at the batch's sample rates; the other streams
// are untouched.  The filter length (AspAecBatch_enable_delay_correction) stays the batch's.
int AspAecBatch_InitStream(AspAecBatch* b, int stream) {
  AspDeviceScope dev_scope_;
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (b->initFlag != kInitCheck) {
    b->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (stream < 0 || stream >= b->S) return aec_fail(ASP_ERR_PARAM, "InitStream: stream out of range");
  AEC_TRY(hipSetDevice(b->device));
  {
    const int rc = enter_vmode(b);
    if (rc != 0) return rc;
  }
  AEC_TRY(hipStreamSynchronize(b->stream));
  {
    StreamCtlScope sc(b, stream);
    const int ext = b->extended, np = b->num_part, nlp = b->nlp_mode;
    const int fs = b->sampFreq, sc_fs = b->scSampFreq;
    init_control(b, fs, sc_fs);
    b->extended = ext;  // the state blocks of a batch have one length
    b->num_part = np;
    b->nlp_mode = nlp;  // (set_config is batch-wide here; Init's own default is the moderate mode the batch started with)
    const int kStateDwords = state_dwords(b);
    std::vector<AspAecState> s0(1);
    init_canonical(s0.data());
    std::vector<float> blk(kStateDwords);
    pack_stream(b, s0.data(), blk.data());
    AEC_TRY(hipMemcpy(b->state + (size_t)stream * kStateDwords, blk.data(), (size_t)kStateDwords * sizeof(float), hipMemcpyHostToDevice));
  }
  // WebRtc_InitBuffer zeroes the far rings: the stream's slice of every slot
  AEC_TRY(hipMemset2D(b->far_ring + (size_t)stream * kFarSlotDwords, (size_t)b->S * kFarSlotDwords * sizeof(float), 0,
                      (size_t)kFarSlotDwords * sizeof(float), (size_t)kFarSlots));
  return 0;
}

static void fill_control(const AspAecBatch* b, AspAecControl* c);

int AspAecBatch_GetControlStream(AspAecBatch* b, int stream, AspAecControl* c) {
  if (!b || !c || stream < 0 || stream >= b->S) return aec_fail(ASP_ERR_PARAM, "GetControlStream: bad argument");
  StreamCtlScope sc(b, stream);
  fill_control(b, c);
  return ASP_OK;
}

static int process_impl(AspAecBatch* b, const float* nearend, const float* near_high, float* out,
                        float* out_high, int nrOfSamples, int msInSndCardBuf, int mem, int32_t skew) {
  if (b && out == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  const int chk = check_running(b, nearend, nrOfSamples);
  if (chk != 0) return chk;
  if (b->sim) {
    int rc_sim = 0;
    b->cur_near_high = near_high;
    b->cur_out_high = out_high;
    const int err_sim = process_device(b, nearend, out, nrOfSamples, msInSndCardBuf, &rc_sim, skew);
    return err_sim != 0 ? err_sim : rc_sim;
  }
  AEC_TRY(hipSetDevice(b->device));
  const size_t bytes = (size_t)b->S * nrOfSamples * sizeof(float);
  const float* nd = nearend;
  float* od = out;
  b->cur_near_high = near_high;
  b->cur_out_high = out_high;
  if (mem == ASP_MEM_HOST) {
    AEC_TRY(hipMemcpyAsync(b->stage_near, nearend, bytes, hipMemcpyHostToDevice, b->stream));
    nd = b->stage_near;
    od = b->stage_out;
    if (b->num_high > 0) {
      AEC_TRY(hipMemcpyAsync(b->stage_near_h, near_high, bytes, hipMemcpyHostToDevice, b->stream));
      b->cur_near_high = b->stage_near_h;
      b->cur_out_high = b->stage_out_h;
    }
  }
  int rc = 0;
  int err;
  if (vmode(b)) {  // the batch's streams have their own control planes: the same delay for every one of them
    const int16_t ms16 = (int16_t)(msInSndCardBuf < -32768 ? -32768 : msInSndCardBuf > 32767 ? 32767 : msInSndCardBuf);
    err = process_v(b, nd, od, nrOfSamples, &ms16, 0, nullptr, &rc);
  } else {
    err = process_device(b, nd, od, nrOfSamples, msInSndCardBuf, &rc, skew);
  }
  if (err != 0) return err;
  if (mem == ASP_MEM_HOST) {
    AEC_TRY(hipMemcpyAsync(out, od, bytes, hipMemcpyDeviceToHost, b->stream));
    if (b->num_high > 0)
      AEC_TRY(hipMemcpyAsync(out_high, b->stage_out_h, bytes, hipMemcpyDeviceToHost, b->stream));
    AEC_TRY(hipStreamSynchronize(b->stream));
  }
  return rc;
}

int AspAecBatch_Run(AspAecBatch* b, const float* farend, const float* nearend, float* out,
                    int nrOfSamples, int num_frames, int msInSndCardBuf, int mem) {
  AspDeviceScope dev_scope_;
  if (b && b->num_high > 0) return aec_fail(ASP_ERR_STATE, "AspAecBatch_Run: single-band batches only (use ProcessBands)");
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_Run: control-only handle");
  if (b && out == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  int chk = check_running(b, farend, nrOfSamples);
  if (chk == 0) chk = check_running(b, nearend, nrOfSamples);
  if (chk != 0) return chk;
  if (num_frames < 0) return aec_fail(ASP_ERR_PARAM, "Run: num_frames < 0");
  if (b->skewMode == kAecTrue)  // the skew estimate is a function of the calls' skew arguments: Process / ProcessBands carry one
    return aec_fail(ASP_ERR_STATE, "Run: no skew argument; with skew compensation on, feed the frames through BufferFarend / Process");
  AEC_TRY(hipSetDevice(b->device));
  const size_t per = (size_t)b->S * nrOfSamples;
  int rc_all = 0;
  if (mem == ASP_MEM_HOST) {
    // stream the frames through device staging in chunks of up to 64 frames
    const int chunk = 64;
    float *dfar = nullptr, *dnear = nullptr, *dout = nullptr;
    AEC_TRY(hipMalloc((void**)&dfar, per * chunk * sizeof(float)));
    hipError_t e = hipMalloc((void**)&dnear, per * chunk * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&dout, per * chunk * sizeof(float));
    int err = e == hipSuccess ? 0 : aec_fail(ASP_ERR_HIP, "Run: staging", e);
    for (int f0 = 0; err == 0 && f0 < num_frames; f0 += chunk) {
      const int nf = num_frames - f0 < chunk ? num_frames - f0 : chunk;
      e = hipMemcpyAsync(dfar, farend + per * f0, per * nf * sizeof(float), hipMemcpyHostToDevice, b->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(dnear, nearend + per * f0, per * nf * sizeof(float), hipMemcpyHostToDevice, b->stream);
      if (e != hipSuccess) {
        err = aec_fail(ASP_ERR_HIP, "Run: upload", e);
        break;
      }
      if (aec_flow_applies(b, nf) && aec_flow_resources(b) == 0) b->flow_rec = true;
      for (int f = 0; f < nf && err == 0; ++f) {
        int rc = 0;
        err = frame_device(b, dfar + per * f, dnear + per * f, dout + per * f, nrOfSamples, msInSndCardBuf, &rc);
        rc_all |= rc;
      }
      if (b->flow_rec) {
        b->flow_rec = false;
        if (err == 0) err = aec_flow_flush(b);
        b->flow_n = 0;
        b->flow_blocks = 0;
      }
      if (err == 0) {
        e = hipMemcpyAsync(out + per * f0, dout, per * nf * sizeof(float), hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) err = aec_fail(ASP_ERR_HIP, "Run: download", e);
      }
    }
    (void)hipStreamSynchronize(b->stream);
    if (err == 0) err = aec_flow_check(b);
    if (dfar) (void)hipFree(dfar);
    if (dnear) (void)hipFree(dnear);
    if (dout) (void)hipFree(dout);
    if (err != 0) return err;
    return rc_all;
  }
  if (aec_flow_applies(b, num_frames)) {
    const int rcf = aec_flow_resources(b);
    if (rcf != 0) return rcf;
    b->flow_rec = true;
  }
  for (int f = 0; f < num_frames; ++f) {
    int rc = 0;
    const int err = frame_device(b, farend + per * f, nearend + per * f, out + per * f, nrOfSamples, msInSndCardBuf, &rc);
    if (err != 0) {
      b->flow_rec = false;
      b->flow_n = 0;
      b->flow_blocks = 0;
      return err;
    }
    rc_all |= rc;
  }
  if (b->flow_rec) {
    b->flow_rec = false;
    const int rcf = aec_flow_flush(b);
    if (rcf != 0) return rcf;
  }
  return rc_all;
}

int AspAecBatch_TimedSteps(AspAecBatch* b, const float* farend, const float* nearend, float* out,
                           int nrOfSamples, int frames_in_ring, int steps, float* elapsed_ms) {
  AspDeviceScope dev_scope_;
  if (b && b->num_high > 0) return aec_fail(ASP_ERR_STATE, "AspAecBatch_TimedSteps: single-band batches only (use ProcessBands)");
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_TimedSteps: control-only handle");
  if (!b || !farend || !nearend || !out || frames_in_ring <= 0 || steps < 0 || !elapsed_ms)
    return aec_fail(ASP_ERR_PARAM, "TimedSteps: bad argument");
  if (check_running(b, farend, nrOfSamples) != 0) return -1;
  if (b->skewMode == kAecTrue)
    return aec_fail(ASP_ERR_STATE, "TimedSteps: no skew argument; with skew compensation on, feed the frames through BufferFarend / Process");
  AEC_TRY(hipSetDevice(b->device));
  const size_t per = (size_t)b->S * nrOfSamples;
  // two chains when the batch is large enough to fill the chip twice over and past its start-up phase
  // (whose pass-through copies stay on the main stream); ASP_AEC_CHAINS=1 keeps one
  const char* ch = getenv("ASP_AEC_CHAINS");
  const bool flow = aec_flow_applies(b, steps);
  if (flow) {
    const int rcf = aec_flow_resources(b);
    if (rcf != 0) return rcf;
  }
  const bool dual = !flow && b->per.empty() && b->S >= 2048 && !b->startup_phase && !(ch && atoi(ch) == 1) && !b->delay_logging && b->reported_delay_enabled && !b->skewMode;
  if (dual && !b->side) {
    AEC_TRY(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
    AEC_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
    AEC_TRY(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
  }
  AEC_TRY(hipEventRecord(b->ev0, b->stream));
  if (dual) {
    AEC_TRY(hipEventRecord(b->ev_fork, b->stream));
    AEC_TRY(hipStreamWaitEvent(b->side, b->ev_fork, 0));
    b->dual = true;
  }
  int err = 0;
  b->flow_rec = flow;
  for (int k = 0; k < steps && err == 0; ++k) {
    const size_t off = per * (size_t)(k % frames_in_ring);
    int rc = 0;
    err = frame_device(b, farend + off, nearend + off, out + off, nrOfSamples, 0, &rc);
  }
  if (flow) {
    b->flow_rec = false;
    if (err == 0) err = aec_flow_flush(b);
    b->flow_n = 0;
    b->flow_blocks = 0;
  }
  if (dual) {
    b->dual = false;
    AEC_TRY(hipEventRecord(b->ev_join, b->side));
    AEC_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
  }
  if (err != 0) return err;
  AEC_TRY(hipEventRecord(b->ev1, b->stream));
  AEC_TRY(hipEventSynchronize(b->ev1));
  AEC_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
  return aec_flow_check(b);
}

int AspAecBatch_Synchronize(AspAecBatch* b) {
  AspDeviceScope dev_scope_;
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_Synchronize: control-only handle");
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  return aec_flow_check(b);
}

int AspAecBatch_SetFlow(AspAecBatch* b, int mode) {
  AspDeviceScope dev_scope_;
  if (!b || mode < -1 || mode > 1) return aec_fail(ASP_ERR_PARAM, "SetFlow: -1 (default), 0 (off) or 1 (on)");
  b->flow = mode;
  return ASP_OK;
}

int AspAecBatch_ExportState(AspAecBatch* b, int stream, AspAecState* out) {
  AspDeviceScope dev_scope_;
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_ExportState: control-only handle");
  if (!b || !out || stream < 0 || stream >= b->S) return aec_fail(ASP_ERR_PARAM, "ExportState: bad argument");
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  const int kStateDwords = state_dwords(b);
  std::vector<float> blk(kStateDwords);
  AEC_TRY(hipMemcpy(blk.data(), b->state + (size_t)stream * kStateDwords, kStateDwords * sizeof(float), hipMemcpyDeviceToHost));
  StreamCtlScope sc(b, stream);  // the partition positions of the block are the stream's own
  unpack_stream(b, blk.data(), out);
  return ASP_OK;
}

int AspAecBatch_ImportState(AspAecBatch* b, int stream, const AspAecState* in) {
  AspDeviceScope dev_scope_;
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_ImportState: control-only handle");
  if (!b || !in || stream < 0 || stream >= b->S) return aec_fail(ASP_ERR_PARAM, "ImportState: bad argument");
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  const int kStateDwords = state_dwords(b);
  std::vector<float> blk(kStateDwords), cur(kStateDwords);
  AEC_TRY(hipMemcpy(cur.data(), b->state + (size_t)stream * kStateDwords, kStateDwords * sizeof(float), hipMemcpyDeviceToHost));
  StreamCtlScope sc(b, stream);
  pack_stream(b, in, blk.data());
  // the time-domain rings are not part of AspAecState: keep the stream's own
  keep_rings(blk.data(), cur.data());
  AEC_TRY(hipMemcpy(b->state + (size_t)stream * kStateDwords, blk.data(), kStateDwords * sizeof(float), hipMemcpyHostToDevice));
  return ASP_OK;
}

// WebRtcAec_enable_delay_correction(WebRtcAec_aec_core(inst), enable) for every stream of the batch
// (aec_core.c:1876-1881): 32 partitions and the extended constants; WebRtcAec_Process then takes the
// ProcessExtended path (echo_cancellation.c:377-394).  Like the reference call it belongs right after Init; used
// mid-stream, the streams' states are carried over (re-packed into blocks of the new size) and behave as the
// reference's do, except that the far-spectrum and filter partitions 12..31 are dropped when the filter is
// shortened (the reference leaves them in place, stale, for a later re-enable).
int AspAecBatch_enable_delay_correction(AspAecBatch* b, int enable) {
  AspDeviceScope dev_scope_;
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  const int np = enable ? kNumPartMax : kNumPartNormal;
  if (!b->per.empty() && np != b->num_part)
    return aec_fail(ASP_ERR_STATE, "enable_delay_correction: the streams have their own control planes; Init the batch first");
  if (np != b->num_part) {
    if (b->xf_pos >= np) {
      // the reference keeps running with the large block position (aec_core.c:1876-1881 stores the flag only); the
      // 12-partition state block has no rows 12..31, so the switch is refused until the position is below 12
      // (include/asp_aec.h); the error code tells a caller of the void drop-in wrapper
      b->lastError = AEC_UNSUPPORTED_FUNCTION_ERROR;
      return aec_fail(ASP_ERR_STATE, "enable_delay_correction(0): xfBufBlockPos is beyond the 12-partition filter; feed frames until it is below 12");
    }
    if (b->far_pending) {
      const int rc = flush_pending_farend(b);
      if (rc != 0) return rc;
    }
    if (!b->sim) {
      AEC_TRY(hipSetDevice(b->device));
      AEC_TRY(hipStreamSynchronize(b->stream));
      const int od = state_dwords(b), nd = AecRows(np).state_dwords;
      std::vector<float> old_all((size_t)b->S * od), new_all((size_t)b->S * nd);
      AEC_TRY(hipMemcpy(old_all.data(), b->state, old_all.size() * sizeof(float), hipMemcpyDeviceToHost));
      std::vector<AspAecState> canon(1);
      const int old_np = b->num_part;
      for (int s = 0; s < b->S; ++s) {
        b->num_part = old_np;
        unpack_stream(b, old_all.data() + (size_t)s * od, canon.data());
        b->num_part = np;
        pack_stream(b, canon.data(), new_all.data() + (size_t)s * nd);
        keep_rings(new_all.data() + (size_t)s * nd, old_all.data() + (size_t)s * od);
      }
      AEC_TRY(hipFree(b->state));
      b->state = nullptr;
      b->num_part = np;
      AEC_TRY(hipMalloc((void**)&b->state, state_bytes(b)));
      AEC_TRY(hipMemcpy(b->state, new_all.data(), new_all.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    b->num_part = np;
  }
  b->extended = enable;  // the reference stores the argument as given
  if (!b->sim) {  // WebRtc_set_allowed_offset(delay_estimator, num_partitions / 2), aec_core.c:1880
    std::vector<int32_t> v((size_t)b->S, b->num_part / 2);
    AEC_TRY(hipSetDevice(b->device));
    AEC_TRY(hipStreamSynchronize(b->stream));
    AEC_TRY(hipMemcpy2D(reinterpret_cast<char*>(b->dblocks) + offsetof(DelayBlock, s.allowed_offset), sizeof(DelayBlock), v.data(),
                        sizeof(int32_t), sizeof(int32_t), (size_t)b->S, hipMemcpyHostToDevice));
  }
  return 0;
}

// WebRtcAec_enable_reported_delay(WebRtcAec_aec_core(inst), enable) for every stream (aec_core.c:1868-1874).
// enable = 0 is the delay-agnostic mode: EstBufDelay is skipped and WebRtcAec_ProcessFrames steers the far-end read
// pointer by the delay estimator (which runs when delay logging is on) instead of the reported delay -- per stream,
// on the device.  Once streams have moved apart, switching the reported delays back on needs a new Init.
int AspAecBatch_enable_reported_delay(AspAecBatch* b, int enable) {
  AspDeviceScope dev_scope_;
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (enable && !b->reported_delay_enabled && b->agn_synced)
    return aec_fail(ASP_ERR_STATE, "enable_reported_delay: the streams' far buffers have moved apart; Init first");
  // the delay-agnostic mode steers every stream's far buffer on the device: a control-only handle refuses it here,
  // not in the middle of a later Process call with its ring positions already advanced
  if (!enable && b->sim) return aec_fail(ASP_ERR_STATE, "enable_reported_delay(0): the delay-agnostic mode needs the device");
  if (!enable && !b->per.empty())
    return aec_fail(ASP_ERR_STATE, "enable_reported_delay(0): per-stream control covers the reported-delay mode; Init the batch first");
  b->reported_delay_enabled = enable ? 1 : 0;
  return 0;
}
int AspAecBatch_reported_delay_enabled(const AspAecBatch* b) { return b ? b->reported_delay_enabled : 0; }

// The delay estimator's state of one stream with the stream's far-buffer read side and system delay.
int AspAecBatch_ExportDelayState(AspAecBatch* b, int stream, AspAecDelayState* out) {
  AspDeviceScope dev_scope_;
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_ExportDelayState: control-only handle");
  if (!b || !out || stream < 0 || stream >= b->S) return aec_fail(ASP_ERR_PARAM, "ExportDelayState: bad argument");
  AEC_TRY(hipSetDevice(b->device));
  if (b->agn_synced && b->nevents > 0) {  // far-end calls the streams' read sides have not seen yet
    const int rc = flush_far_events(b);
    if (rc != 0) return rc;
  }
  AEC_TRY(hipStreamSynchronize(b->stream));
  AEC_TRY(hipMemcpy(out, &b->dblocks[stream].s, sizeof *out, hipMemcpyDeviceToHost));
  if (!b->agn_synced) {  // lock-step: the batch's values
    out->far_read = b->far_pos.read;
    out->far_write = b->far_pos.write;
    out->far_wrap = b->far_pos.wrap;
    out->system_delay = b->system_delay;
  }
  return ASP_OK;
}

// WebRtcAec_GetDelayMetrics for every stream (echo_cancellation.c:550-571, aec_core.c:1780-1836): median and
// spread (L1 norm around the median) of the block-wise delay estimates since the last call, in ms; the
// histograms are cleared.  median / std [num_streams].
int AspAecBatch_GetDelayMetrics(AspAecBatch* b, int* median, int* std) {
  AspDeviceScope dev_scope_;
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_GetDelayMetrics: control-only handle");
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (median == nullptr || std == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (b->initFlag != kInitCheck) {
    b->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  if (b->delay_logging == 0) {
    b->lastError = AEC_UNSUPPORTED_FUNCTION_ERROR;  // logging disabled
    return -1;
  }
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  constexpr int H = ASP_AEC_DELAY_HISTORY;
  std::vector<int32_t> hist((size_t)b->S * H), look((size_t)b->S);
  AEC_TRY(hipMemcpy2D(hist.data(), H * sizeof(int32_t), reinterpret_cast<char*>(b->dblocks) + offsetof(DelayBlock, s.delay_histogram),
                      sizeof(DelayBlock), H * sizeof(int32_t), (size_t)b->S, hipMemcpyDeviceToHost));
  AEC_TRY(hipMemcpy2D(look.data(), sizeof(int32_t), reinterpret_cast<char*>(b->dblocks) + offsetof(DelayBlock, s.lookahead),
                      sizeof(DelayBlock), sizeof(int32_t), (size_t)b->S, hipMemcpyDeviceToHost));
  const int kMsPerBlock = kPartLen / (b->mult * 8);
  for (int s = 0; s < b->S; ++s) {
    const int32_t* h = hist.data() + (size_t)s * H;
    int num_delay_values = 0, my_median = 0;
    for (int i = 0; i < H; i++) num_delay_values += h[i];
    if (num_delay_values == 0) {
      median[s] = -1;
      std[s] = -1;
      continue;
    }
    int delay_values = num_delay_values >> 1;
    for (int i = 0; i < H; i++) {
      delay_values -= h[i];
      if (delay_values < 0) {
        my_median = i;
        break;
      }
    }
    median[s] = (my_median - look[s]) * kMsPerBlock;
    float l1_norm = 0;
    for (int i = 0; i < H; i++) l1_norm += (float)abs(i - my_median) * h[i];
    std[s] = (int)(l1_norm / (float)num_delay_values + 0.5f) * kMsPerBlock;
  }
  // (a stream without values keeps its -- empty -- histogram; the others are cleared: aec_core.c:1833)
  AEC_TRY(hipMemset2DAsync(reinterpret_cast<char*>(b->dblocks) + offsetof(DelayBlock, s.delay_histogram), sizeof(DelayBlock), 0,
                           H * sizeof(int32_t), (size_t)b->S, b->stream));
  AEC_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

int AspAecBatch_delay_correction_enabled(const AspAecBatch* b) { return b ? b->extended : 0; }

int AspAecBatch_GetControl(const AspAecBatch* b, AspAecControl* c) {
  if (!b || !c) return aec_fail(ASP_ERR_PARAM, "GetControl: bad argument");
  fill_control(b, c);  // (with per-stream control: stream 0's)
  return ASP_OK;
}

static void fill_control(const AspAecBatch* b, AspAecControl* c) {
  c->startup_phase = b->startup_phase;
  c->checkBuffSize = b->checkBuffSize;
  c->bufSizeStart = b->bufSizeStart;
  c->knownDelay = b->knownDelay;
  c->filtDelay = b->filtDelay;
  c->timeForDelayChange = b->timeForDelayChange;
  c->lastDelayDiff = b->lastDelayDiff;
  c->counter = b->counter;
  c->sum = b->sum;
  c->firstVal = b->firstVal;
  c->checkBufSizeCtr = b->checkBufSizeCtr;
  c->system_delay = b->system_delay;
  c->core_knownDelay = b->core_knownDelay;
  c->far_read = b->far_pos.read;
  c->far_write = b->far_pos.write;
  c->far_wrap = b->far_pos.wrap;
  c->pre_read = b->pre_pos.read;
  c->pre_write = b->pre_pos.write;
  c->pre_wrap = b->pre_pos.wrap;
  c->near_read = b->near_pos.read;
  c->near_write = b->near_pos.write;
  c->near_wrap = b->near_pos.wrap;
  c->out_read = b->out_pos.read;
  c->out_write = b->out_pos.write;
  c->out_wrap = b->out_pos.wrap;
  c->blocks_processed = b->blocks_processed;
}

int AspAecBatch_get_echo_status(AspAecBatch* b, int* status) {
  AspDeviceScope dev_scope_;
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_get_echo_status: control-only handle");
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (status == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (b->initFlag != kInitCheck) {
    b->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  for (int s = 0; s < b->S; ++s) {
    int32_t v = 0;
    AEC_TRY(hipMemcpy(&v, b->state + (size_t)s * state_dwords(b) + kOffScalars + S_ECHOSTATE, sizeof v, hipMemcpyDeviceToHost));
    status[s] = v;
  }
  return 0;
}

namespace {
void level_of(const AspAecStats& s, AecLevel* out) {  // echo_cancellation.c:484-499 (and the erle / aNlp copies)
  const float kUpWeight = 0.7f;
  out->instant = (int)s.instant;
  if ((s.himean > -100) && (s.average > -100)) {
    const float dtmp = kUpWeight * s.himean + (1 - kUpWeight) * s.average;
    out->average = (int)dtmp;
  } else {
    out->average = -100;
  }
  out->max = (int)s.max;
  out->min = s.min < 100 ? (int)s.min : -100;
}
}  // namespace

int AspAecBatch_GetMetrics(AspAecBatch* b, AecMetrics* out) {  // WebRtcAec_GetMetrics, echo_cancellation.c:456-548
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_GetMetrics: control-only handle");
  if (!b) return aec_fail(ASP_ERR_PARAM, "null batch handle");
  if (out == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (b->initFlag != kInitCheck) {
    b->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  std::vector<float> all((size_t)b->S * kMetDwords);
  AEC_TRY(hipMemcpy(all.data(), b->metrics, all.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int s = 0; s < b->S; ++s) {
    AspAecMetricsState m;
    memcpy(&m, all.data() + (size_t)s * kMetDwords, sizeof m);
    AecMetrics* o = out + s;
    level_of(m.erl, &o->erl);
    level_of(m.erle, &o->erle);
    const int stmp = (o->erl.average > -100 && o->erle.average > -100) ? o->erl.average + o->erle.average : -100;
    o->rerl.average = stmp;
    o->rerl.instant = stmp;
    o->rerl.max = stmp;
    o->rerl.min = stmp;
    level_of(m.aNlp, &o->aNlp);
  }
  return 0;
}

int AspAecBatch_ExportMetricsState(AspAecBatch* b, int stream, AspAecMetricsState* out) {
  AspDeviceScope dev_scope_;
  if (b && b->sim) return aec_fail(ASP_ERR_STATE, "AspAecBatch_ExportMetricsState: control-only handle");
  if (!b || !out || stream < 0 || stream >= b->S) return aec_fail(ASP_ERR_PARAM, "ExportMetricsState: bad argument");
  AEC_TRY(hipSetDevice(b->device));
  AEC_TRY(hipStreamSynchronize(b->stream));
  AEC_TRY(hipMemcpy(out, b->metrics + (size_t)stream * kMetDwords, sizeof *out, hipMemcpyDeviceToHost));
  return 0;
}

// Diagnostic: one BufferFarend + Process on device frames with phase time stamps (s_memtime
// ticks) of stream 0's first block.
int AspAecBatch_DebugStamps(AspAecBatch* b, const float* far_dev, const float* near_dev, float* out_dev,
                            unsigned long long* stamps16) {
  AspDeviceScope dev_scope_;
  if (!b || !far_dev || !near_dev || !out_dev || !stamps16) return aec_fail(ASP_ERR_PARAM, "DebugStamps: bad argument");
  AEC_TRY(hipSetDevice(b->device));
  unsigned long long* d = nullptr;
  AEC_TRY(hipMalloc((void**)&d, 16 * sizeof(unsigned long long)));
  hipError_t e = hipMemset(d, 0, 16 * sizeof(unsigned long long));
  int err = 0, rc = 0;
  if (e == hipSuccess) {
    b->debug_stamps = d;
    err = buffer_farend_device(b, far_dev, 160);
    if (err == 0) err = process_device(b, near_dev, out_dev, 160, 0, &rc);
    b->debug_stamps = nullptr;
    e = hipStreamSynchronize(b->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(stamps16, d, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return aec_fail(ASP_ERR_HIP, "DebugStamps", e);
  return err;
}

int AspAec_rdft128_batch(const float* src, float* dst, int isgn, int count, int device) {
  AspDeviceScope dev_scope_;
  if (!src || !dst || count <= 0) return aec_fail(ASP_ERR_PARAM, "rdft128_batch: bad argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return aec_fail(ASP_ERR_NO_DEVICE, "no HIP device: the echo canceller has no CPU fallback");
  AEC_TRY(hipSetDevice(device));
  float *d_in = nullptr, *d_out = nullptr;
  AecTables* d_t = nullptr;
  const size_t bytes = (size_t)count * 128 * sizeof(float);
  AecTables T;
  build_tables(&T);
  hipError_t e = hipMalloc((void**)&d_in, bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&d_out, bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&d_t, sizeof T);
  if (e == hipSuccess) e = hipMemcpy(d_t, &T, sizeof T, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_in, src, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_aec_rdft128(d_in, d_out, isgn, count, d_t, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(dst, d_out, bytes, hipMemcpyDeviceToHost);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_t) (void)hipFree(d_t);
  if (e != hipSuccess) return aec_fail(ASP_ERR_HIP, "rdft128_batch", e);
  return ASP_OK;
}

int AspAec_delay_estimator_batch(AspAecDelayState* states, int count, const uint32_t* binary_far, const uint32_t* binary_near,
                                 int nblocks, int device) {
  AspDeviceScope dev_scope_;
  if (!states || !binary_far || !binary_near || count <= 0 || nblocks < 0)
    return aec_fail(ASP_ERR_PARAM, "delay_estimator_batch: bad argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return aec_fail(ASP_ERR_NO_DEVICE, "no HIP device: the echo canceller has no CPU fallback");
  AEC_TRY(hipSetDevice(device));
  std::vector<DelayBlock> blocks((size_t)count);
  for (int i = 0; i < count; ++i) {
    memset(&blocks[i], 0, sizeof(DelayBlock));
    blocks[i].s = states[i];
  }
  std::vector<unsigned> words((size_t)count * kFlowBitsBlocks * 2);
  DelayBlock* d_blocks = nullptr;
  unsigned* d_bits = nullptr;
  hipError_t e = hipMalloc((void**)&d_blocks, blocks.size() * sizeof(DelayBlock));
  if (e == hipSuccess) e = hipMalloc((void**)&d_bits, words.size() * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemcpy(d_blocks, blocks.data(), blocks.size() * sizeof(DelayBlock), hipMemcpyHostToDevice);
  for (int k0 = 0; k0 < nblocks && e == hipSuccess; k0 += kFlowBitsBlocks) {  // as many launches as hand-off launches would carry
    const int nb = nblocks - k0 < kFlowBitsBlocks ? nblocks - k0 : kFlowBitsBlocks;
    for (int i = 0; i < count; ++i)
      for (int k = 0; k < nb; ++k) {
        words[((size_t)i * kFlowBitsBlocks + k) * 2] = binary_far[(size_t)i * nblocks + k0 + k];
        words[((size_t)i * kFlowBitsBlocks + k) * 2 + 1] = binary_near[(size_t)i * nblocks + k0 + k];
      }
    e = hipMemcpy(d_bits, words.data(), words.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_aec_delay_bits(d_blocks, d_bits, count, nb, 1, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(blocks.data(), d_blocks, blocks.size() * sizeof(DelayBlock), hipMemcpyDeviceToHost);
  if (d_blocks) (void)hipFree(d_blocks);
  if (d_bits) (void)hipFree(d_bits);
  if (e != hipSuccess) return aec_fail(ASP_ERR_HIP, "delay_estimator_batch", e);
  for (int i = 0; i < count; ++i) states[i] = blocks[i].s;
  return ASP_OK;
}

int AspAec_host_table(int which, float* out, int capacity) {
  AspDeviceScope dev_scope_;
  AecTables T;
  build_tables(&T);
  const float* src = nullptr;
  int n = 0;
  switch (which) {
    case 0: src = T.w; n = 64; break;
    case 1: src = T.wk3a; n = 16; break;
    case 2: src = T.wk3b; n = 16; break;
    case 3: src = T.hann; n = 65; break;
    case 4: src = T.weight; n = 65; break;
    case 5: src = T.odrive; n = 65; break;
    default: return aec_fail(ASP_ERR_PARAM, "host_table: unknown table");
  }
  if (!out || capacity < n) return aec_fail(ASP_ERR_PARAM, "host_table: buffer too small");
  memcpy(out, src, n * sizeof(float));
  return n;
}

// ------------------------------------------------------------------ layer 1
// The reference's per-stream API: a handle is a batch of one stream, host buffers.
int32_t WebRtcAec_Create(void** aecInst) {  // echo_cancellation.c:121-168
  if (aecInst == nullptr) return -1;
  AspAecBatch* b = nullptr;
  if (AspAecBatch_Create(&b, 1, 0) != ASP_OK) {
    *aecInst = nullptr;
    return -1;
  }
  *aecInst = b;
  return 0;
}

int32_t WebRtcAec_Free(void* aecInst) {
  AspDeviceScope dev_scope_;
  if (aecInst == nullptr) return -1;
  return AspAecBatch_Free((AspAecBatch*)aecInst);
}

int32_t WebRtcAec_Init(void* aecInst, int32_t sampFreq, int32_t scSampFreq) {
  AspDeviceScope dev_scope_;
  return AspAecBatch_Init((AspAecBatch*)aecInst, sampFreq, scSampFreq) == 0 ? 0 : -1;
}

int32_t WebRtcAec_BufferFarend(void* aecInst, const float* farend, int16_t nrOfSamples) {
  AspDeviceScope dev_scope_;
  return AspAecBatch_BufferFarend((AspAecBatch*)aecInst, farend, nrOfSamples, ASP_MEM_HOST) == 0 ? 0 : -1;
}

int32_t WebRtcAec_Process(void* aecInst, const float* const* nearend, int num_bands,
                          float* const* out, int16_t nrOfSamples, int16_t msInSndCardBuf,
                          int32_t skew) {
  AspDeviceScope dev_scope_;
  AspAecBatch* b = (AspAecBatch*)aecInst;
  if (b == nullptr) return -1;
  if (out == nullptr || nearend == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (num_bands != 1 + b->num_high) {  // the reference asserts aec->num_bands == num_bands (aec_core.c:1683)
    b->lastError = AEC_BAD_PARAMETER_ERROR;
    return -1;
  }
  if (num_bands == 2)
    return AspAecBatch_ProcessBands(b, nearend[0], nearend[1], out[0], out[1], nrOfSamples, msInSndCardBuf,
                                    skew, ASP_MEM_HOST) == 0 ? 0 : -1;
  return AspAecBatch_Process(b, nearend[0], out[0], nrOfSamples, msInSndCardBuf, skew, ASP_MEM_HOST) == 0 ? 0 : -1;
}

int WebRtcAec_set_config(void* handle, AecConfig config) {
  AspDeviceScope dev_scope_;
  return AspAecBatch_set_config((AspAecBatch*)handle, config) == 0 ? 0 : -1;
}

int WebRtcAec_get_echo_status(void* handle, int* status) {
  AspDeviceScope dev_scope_;
  return AspAecBatch_get_echo_status((AspAecBatch*)handle, status) == 0 ? 0 : -1;
}

int WebRtcAec_GetMetrics(void* handle, AecMetrics* metrics) {  // echo_cancellation.c:456-548
  AspAecBatch* b = (AspAecBatch*)handle;
  if (b == nullptr) return -1;
  if (metrics == nullptr) {
    b->lastError = AEC_NULL_POINTER_ERROR;
    return -1;
  }
  if (b->initFlag != kInitCheck) {
    b->lastError = AEC_UNINITIALIZED_ERROR;
    return -1;
  }
  return AspAecBatch_GetMetrics(b, metrics) == 0 ? 0 : -1;
}

int WebRtcAec_GetDelayMetrics(void* handle, int* median, int* std) {  // echo_cancellation.c:550-571
  AspAecBatch* b = (AspAecBatch*)handle;
  if (b == nullptr) return -1;
  return AspAecBatch_GetDelayMetrics(b, median, std) == 0 ? 0 : -1;
}

int32_t WebRtcAec_get_error_code(void* aecInst) {
  AspDeviceScope dev_scope_;
  return AspAecBatch_get_error_code((AspAecBatch*)aecInst);
}

struct AecCore* WebRtcAec_aec_core(void* handle) {
  AspDeviceScope dev_scope_;
  return reinterpret_cast<struct AecCore*>(handle);
}

// aec_core.h:129-133 / aec_core.c:1876-1885: `self` is the token WebRtcAec_aec_core returned
void WebRtcAec_enable_delay_correction(struct AecCore* self, int enable) {
  AspDeviceScope dev_scope_;
  (void)AspAecBatch_enable_delay_correction(reinterpret_cast<AspAecBatch*>(self), enable);
}

int WebRtcAec_delay_correction_enabled(struct AecCore* self) {
  AspDeviceScope dev_scope_;
  return AspAecBatch_delay_correction_enabled(reinterpret_cast<AspAecBatch*>(self));
}

// aec_core.h:110-114 / aec_core.c:1886-1894 (what echo_cancellation_unittest.cc:34-48 and the APM use): the buffered
// far-end delay in samples.  In the delay-agnostic mode it is the stream's own (a handle is a batch of one stream).
int WebRtcAec_system_delay(struct AecCore* self) {
  AspDeviceScope dev_scope_;
  AspAecBatch* b = reinterpret_cast<AspAecBatch*>(self);
  if (!b) return 0;
  if (b->agn_synced && !b->sim) {
    AspAecDelayState d;
    if (AspAecBatch_ExportDelayState(b, 0, &d) == ASP_OK) return d.system_delay;
  }
  return b->system_delay;
}
void WebRtcAec_SetSystemDelay(struct AecCore* self, int delay) {
  AspDeviceScope dev_scope_;
  AspAecBatch* b = reinterpret_cast<AspAecBatch*>(self);
  if (!b || delay < 0) return;  // the reference asserts delay >= 0
  b->system_delay = delay;
  if (b->agn_synced && !b->sim) {
    std::vector<int32_t> v((size_t)b->S, delay);
    if (hipSetDevice(b->device) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) return;
    (void)hipMemcpy2D(reinterpret_cast<char*>(b->dblocks) + offsetof(DelayBlock, s.system_delay), sizeof(DelayBlock), v.data(),
                      sizeof(int32_t), sizeof(int32_t), (size_t)b->S, hipMemcpyHostToDevice);
  }
}

// aec_core.h:121-126 / aec_core.c:1868-1874
void WebRtcAec_enable_reported_delay(struct AecCore* self, int enable) {
  AspDeviceScope dev_scope_;
  (void)AspAecBatch_enable_reported_delay(reinterpret_cast<AspAecBatch*>(self), enable);
}

int WebRtcAec_reported_delay_enabled(struct AecCore* self) {
  AspDeviceScope dev_scope_;
  return AspAecBatch_reported_delay_enabled(reinterpret_cast<AspAecBatch*>(self));
}

}  // extern "C"
