// sinc_api.hip -- host side of include/asp_resample.h: kernel table, the position arithmetic of
// SincResampler::Resample / PushSincResampler::Resample run once per call for the whole batch
// (doubles, exactly the reference's recurrence), and the batch handle.  No CPU fallback.
#include <hip/hip_runtime.h>

#include "device_scope.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "asp_ns.h"
#include "asp_resample.h"
#include "sinc_layout.h"

using namespace aspsinc;

namespace aspsinc {
hipError_t launch_sinc(float* state, const float* kernel_table, const OutDesc* desc,
                       const int16_t* in, int16_t* out, int num_channels, int buf_len,
                       int src_frames, int dst_frames, const SincPlan& plan, hipStream_t s);
}

namespace {
thread_local char g_sinc_err[512] = "";
int sinc_fail(int code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess)
    snprintf(g_sinc_err, sizeof g_sinc_err, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(g_sinc_err, sizeof g_sinc_err, "%s", what);
  fprintf(stderr, "asp_resample: %s\n", g_sinc_err);
  return code;
}
#define SINC_TRY(expr)                                             \
  do {                                                             \
    hipError_t e_ = (expr);                                        \
    if (e_ != hipSuccess) return sinc_fail(ASP_ERR_HIP, #expr, e_); \
  } while (0)
}  // namespace

struct AspSincBatch {
  int C = 0, device = 0, src = 0, dst = 0, buf_len = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  float* state = nullptr;   // [C][buf_len]
  float* ktable = nullptr;  // [33 * 32]
  // descriptor tables: a ring of kDescRing slots on the device, each with a pinned host twin and the
  // event of the last kernel that read it (a call whose table differs from the previous one takes
  // the next slot: no stream synchronisation on the way)
  static constexpr int kDescRing = 4;
  OutDesc* desc = nullptr;         // [kDescRing][max_out]
  OutDesc* desc_pinned = nullptr;  // [kDescRing][max_out], hipHostMalloc
  hipEvent_t desc_ev[kDescRing] = {};
  size_t max_out = 0;
  int desc_slot = 0;
  int16_t *s_in = nullptr, *s_out = nullptr;
  std::vector<float> kernel_host;
  std::vector<OutDesc> desc_dev;  // the descriptor table as last uploaded
  std::vector<OutDesc> desc_host;
  // SincResampler / PushSincResampler position state (identical for every channel)
  double ratio = 0, vsi = 0;
  int r0 = 0, r3 = 0, r4 = 0, block_size = 0;
  bool primed = false, first_pass = true;
};

namespace {

void update_regions(AspSincBatch* b, bool second_load) {  // sinc_resampler.cc:190-199
  b->r0 = second_load ? kKernelSize : kKernelSize / 2;
  b->r3 = b->r0 + b->src - kKernelSize;
  b->r4 = b->r0 + b->src - kKernelSize / 2;
  b->block_size = b->r4 - kKernelSize / 2;
}

// NOTE: C++ translation unit; libm calls take explicit doubles so the arithmetic is the reference's.
void init_kernel(AspSincBatch* b) {  // sinc_resampler.cc:201-232, 88-101
  const double kAlpha = 0.16;
  const double kA0 = 0.5 * (1.0 - kAlpha), kA1 = 0.5, kA2 = 0.5 * kAlpha;
  double sinc_scale_factor = b->ratio > 1.0 ? 1.0 / b->ratio : 1.0;
  sinc_scale_factor *= 0.9;
  b->kernel_host.assign((size_t)kKernelSize * (kKernelOffsetCount + 1), 0.f);
  for (int offset_idx = 0; offset_idx <= kKernelOffsetCount; ++offset_idx) {
    const float subsample_offset = static_cast<float>(offset_idx) / kKernelOffsetCount;
    for (int i = 0; i < kKernelSize; ++i) {
      const int idx = i + offset_idx * kKernelSize;
      const float pre_sinc = static_cast<float>(M_PI * (double)(i - kKernelSize / 2 - subsample_offset));
      const float x = (i - subsample_offset) / kKernelSize;
      const float window = static_cast<float>(kA0 - kA1 * cos(2.0 * M_PI * (double)x) + kA2 * cos(4.0 * M_PI * (double)x));
      b->kernel_host[idx] = static_cast<float>(
          (double)window * ((pre_sinc == 0) ? sinc_scale_factor
                                            : (sin(sinc_scale_factor * (double)pre_sinc) / (double)pre_sinc)));
    }
  }
}

// SincResampler::Resample(frames, destination) as a plan: appends segments / descriptors.
// dest_base >= 0: outputs land at dest_base + k of the caller's buffer; < 0: discarded.
int plan_resample(AspSincBatch* b, int frames, int dest_base, SincPlan* plan) {
  int remaining = frames, produced = 0;
  auto open_seg = [&](int load, int shift) -> SincSeg* {
    if (plan->nseg >= 6) return nullptr;
    SincSeg* sg = &plan->seg[plan->nseg++];
    sg->load = load;
    sg->shift = shift;
    sg->r0 = b->r0;
    sg->r3 = b->r3;
    sg->out_begin = sg->out_end = (int)b->desc_host.size();
    return sg;
  };
  SincSeg* cur = nullptr;
  if (!b->primed && remaining) {  // read_cb_->Run(request_frames_, r0_)
    cur = open_seg(b->first_pass ? 1 : 2, 0);
    if (!cur) return sinc_fail(ASP_ERR_STATE, "resampler plan: too many segments");
    b->first_pass = false;
    b->primed = true;
  }
  if (!cur) {
    cur = open_seg(0, 0);
    if (!cur) return sinc_fail(ASP_ERR_STATE, "resampler plan: too many segments");
  }
  while (remaining) {
    for (int i = (int)ceil((b->block_size - b->vsi) / b->ratio); i > 0; --i) {
      const int source_idx = (int)b->vsi;
      const double subsample_remainder = b->vsi - source_idx;
      const double virtual_offset_idx = subsample_remainder * kKernelOffsetCount;
      const int offset_idx = (int)virtual_offset_idx;
      const double factor = virtual_offset_idx - offset_idx;
      OutDesc d;
      d.source_idx = source_idx;
      d.offset_idx = offset_idx;
      d.f1 = static_cast<float>(1.0 - factor);
      d.f2 = static_cast<float>(factor);
      d.dest = dest_base >= 0 ? dest_base + produced : -1;
      b->desc_host.push_back(d);
      cur->out_end = (int)b->desc_host.size();
      ++produced;
      b->vsi += b->ratio;
      if (!--remaining) return 0;
    }
    b->vsi -= b->block_size;
    const int r3_before = b->r3;
    if (b->r0 == kKernelSize / 2) update_regions(b, true);
    const int load = b->first_pass ? 1 : 2;
    b->first_pass = false;
    cur = open_seg(load, 1);
    if (!cur) return sinc_fail(ASP_ERR_STATE, "resampler plan: too many segments");
    cur->r3 = r3_before;  // the shift reads the region of the block just consumed
  }
  return 0;
}

}  // namespace

extern "C" {

int AspSincBatch_Create(AspSincBatch** out, int num_channels, int source_frames,
                        int destination_frames, int device) {
  AspDeviceScope dev_scope_;
  if (!out || num_channels <= 0 || source_frames <= kKernelSize || destination_frames <= 0)
    return sinc_fail(ASP_ERR_PARAM, "AspSincBatch_Create: bad argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return sinc_fail(ASP_ERR_NO_DEVICE, "no HIP device: the resampler has no CPU fallback");
  if (device < 0 || device >= count) return sinc_fail(ASP_ERR_PARAM, "device ordinal out of range");
  SINC_TRY(hipSetDevice(device));
  AspSincBatch* b = new AspSincBatch();
  b->C = num_channels;
  b->device = device;
  b->src = source_frames;
  b->dst = destination_frames;
  b->buf_len = source_frames + kKernelSize;
  b->ratio = source_frames * 1.0 / destination_frames;  // push_sinc_resampler.cc:19
  b->vsi = 0;                                            // Flush, sinc_resampler.cc:335-341
  update_regions(b, false);
  init_kernel(b);
  const size_t max_out = (size_t)destination_frames * 2 + 64;
  b->max_out = max_out;
  hipError_t e = hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking);
  b->stream = b->own_stream;
  if (e == hipSuccess) e = hipMalloc((void**)&b->state, (size_t)num_channels * b->buf_len * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->ktable, b->kernel_host.size() * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b->desc, AspSincBatch::kDescRing * max_out * sizeof(OutDesc));
  if (e == hipSuccess) e = hipHostMalloc((void**)&b->desc_pinned, AspSincBatch::kDescRing * max_out * sizeof(OutDesc));
  for (int i = 0; i < AspSincBatch::kDescRing && e == hipSuccess; ++i)
    e = hipEventCreateWithFlags(&b->desc_ev[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&b->s_in, (size_t)num_channels * source_frames * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->s_out, (size_t)num_channels * destination_frames * sizeof(int16_t));
  if (e == hipSuccess) e = hipMemsetAsync(b->state, 0, (size_t)num_channels * b->buf_len * sizeof(float), b->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(b->ktable, b->kernel_host.data(), b->kernel_host.size() * sizeof(float),
                       hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e != hipSuccess) {
    AspSincBatch_Free(b);
    return sinc_fail(ASP_ERR_HIP, "AspSincBatch_Create", e);
  }
  *out = b;
  return ASP_OK;
}

int AspSincBatch_Free(AspSincBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return -1;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->state) (void)hipFree(b->state);
  if (b->ktable) (void)hipFree(b->ktable);
  if (b->desc) (void)hipFree(b->desc);
  if (b->desc_pinned) (void)hipHostFree(b->desc_pinned);
  for (int i = 0; i < AspSincBatch::kDescRing; ++i)
    if (b->desc_ev[i]) (void)hipEventDestroy(b->desc_ev[i]);
  if (b->s_in) (void)hipFree(b->s_in);
  if (b->s_out) (void)hipFree(b->s_out);
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  delete b;
  return 0;
}

int AspSincBatch_SetStream(AspSincBatch* b, void* hip_stream) {
  AspDeviceScope dev_scope_;
  if (!b) return sinc_fail(ASP_ERR_PARAM, "null batch handle");
  SINC_TRY(hipSetDevice(b->device));
  SINC_TRY(hipStreamSynchronize(b->stream));
  b->stream = hip_stream ? (hipStream_t)hip_stream : b->own_stream;
  return ASP_OK;
}

int AspSincBatch_num_channels(const AspSincBatch* b) { return b ? b->C : 0; }

int AspSincBatch_Resample(AspSincBatch* b, const int16_t* in, int16_t* out, int mem) {
  AspDeviceScope dev_scope_;
  if (!b || !in || !out) return sinc_fail(ASP_ERR_PARAM, "AspSincBatch_Resample: bad argument");
  SINC_TRY(hipSetDevice(b->device));
  // PushSincResampler::Resample (push_sinc_resampler.cc:47-60): a priming pass on the first call
  SincPlan plan;
  memset(&plan, 0, sizeof plan);
  b->desc_host.clear();
  int rc = 0;
  if (b->first_pass) rc = plan_resample(b, (int)(b->block_size / b->ratio), -1, &plan);  // ChunkSize()
  if (rc == 0) rc = plan_resample(b, b->dst, 0, &plan);
  if (rc != 0) return rc;
  const int16_t* din = in;
  int16_t* dout = out;
  if (mem == ASP_MEM_HOST) {
    SINC_TRY(hipMemcpyAsync(b->s_in, in, (size_t)b->C * b->src * sizeof(int16_t), hipMemcpyHostToDevice, b->stream));
    din = b->s_in;
    dout = b->s_out;
  } else if (mem != ASP_MEM_DEVICE) {
    return sinc_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  // The descriptor table repeats from call to call while the fractional position does: upload it only
  // when it differs from the one last uploaded, into the next slot of the ring (the kernel that read
  // that slot kDescRing uploads ago has long finished; its event says so without draining the stream).
  if (b->desc_host.size() > b->max_out) return sinc_fail(ASP_ERR_STATE, "descriptor table overflow");
  if (b->desc_dev.size() != b->desc_host.size() ||
      memcmp(b->desc_dev.data(), b->desc_host.data(), b->desc_host.size() * sizeof(OutDesc)) != 0) {
    b->desc_slot = (b->desc_slot + 1) % AspSincBatch::kDescRing;
    SINC_TRY(hipEventSynchronize(b->desc_ev[b->desc_slot]));
    OutDesc* pin = b->desc_pinned + (size_t)b->desc_slot * b->max_out;
    memcpy(pin, b->desc_host.data(), b->desc_host.size() * sizeof(OutDesc));
    SINC_TRY(hipMemcpyAsync(b->desc + (size_t)b->desc_slot * b->max_out, pin, b->desc_host.size() * sizeof(OutDesc),
                            hipMemcpyHostToDevice, b->stream));
    b->desc_dev = b->desc_host;
  }
  SINC_TRY(launch_sinc(b->state, b->ktable, b->desc + (size_t)b->desc_slot * b->max_out, din, dout, b->C, b->buf_len,
                       b->src, b->dst, plan, b->stream));
  SINC_TRY(hipEventRecord(b->desc_ev[b->desc_slot], b->stream));
  if (mem == ASP_MEM_HOST) {
    SINC_TRY(hipMemcpyAsync(out, dout, (size_t)b->C * b->dst * sizeof(int16_t), hipMemcpyDeviceToHost, b->stream));
    SINC_TRY(hipStreamSynchronize(b->stream));
  }
  return ASP_OK;
}

int AspSincBatch_Synchronize(AspSincBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return sinc_fail(ASP_ERR_PARAM, "null batch handle");
  SINC_TRY(hipSetDevice(b->device));
  SINC_TRY(hipStreamSynchronize(b->stream));
  return ASP_OK;
}

int AspSincBatch_kernel_table(const AspSincBatch* b, float* out, int capacity) {
  AspDeviceScope dev_scope_;
  if (!b || !out || capacity < (int)b->kernel_host.size()) return sinc_fail(ASP_ERR_PARAM, "kernel_table: bad argument");
  memcpy(out, b->kernel_host.data(), b->kernel_host.size() * sizeof(float));
  return (int)b->kernel_host.size();
}

}  // extern "C"
