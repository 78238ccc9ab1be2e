// qmf_api.hip -- host side of include/asp_split.h: the batch handle (filter states in HBM) and
// the reference's WebRtcSpl_AnalysisQMF / WebRtcSpl_SynthesisQMF as a batch of one channel with
// caller-owned states.  No CPU fallback.
#include <hip/hip_runtime.h>

#include "device_scope.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asp_ns.h"
#include "asp_resample.h"
#include "asp_split.h"

namespace aspqmf {
hipError_t launch_analysis(int32_t* state, const int16_t* in, int16_t* low, int16_t* high,
                           int num_channels, int band_length, hipStream_t s);
hipError_t launch_synthesis(int32_t* state, const int16_t* low, const int16_t* high, int16_t* out,
                            int num_channels, int band_length, hipStream_t s);
hipError_t launch_analysis_pair(int32_t* state0, const int16_t* in0, int16_t* low0, int16_t* high0,
                                int32_t* state1, const int16_t* in1, int16_t* low1, int16_t* high1,
                                int16_t* drop_scratch, int num_channels, int band_length, hipStream_t s);
hipError_t launch_synthesis_pair(int32_t* state0, const int16_t* low0, const int16_t* high0, int16_t* out0,
                                 int32_t* state1, const int16_t* low1, const int16_t* high1, int16_t* out1,
                                 int num_channels, int band_length, hipStream_t s);
}  // namespace aspqmf

namespace {
thread_local char g_qmf_err[512] = "";
int qmf_fail(int code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess)
    snprintf(g_qmf_err, sizeof g_qmf_err, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(g_qmf_err, sizeof g_qmf_err, "%s", what);
  fprintf(stderr, "asp_split: %s\n", g_qmf_err);
  return code;
}
#define QMF_TRY(expr)                                             \
  do {                                                            \
    hipError_t e_ = (expr);                                       \
    if (e_ != hipSuccess) return qmf_fail(ASP_ERR_HIP, #expr, e_); \
  } while (0)
}  // namespace

struct AspQmfBatch {
  int C = 0, device = 0;
  hipStream_t stream = nullptr;
  int32_t* state = nullptr;                  // [C][24]
  int16_t *s_in = nullptr, *s_a = nullptr, *s_b = nullptr;  // staging [C][640], [C][320] x 2
};

extern "C" {

int AspQmfBatch_Create(AspQmfBatch** out, int num_channels, int device) {
  AspDeviceScope dev_scope_;
  if (!out || num_channels <= 0) return qmf_fail(ASP_ERR_PARAM, "AspQmfBatch_Create: bad argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return qmf_fail(ASP_ERR_NO_DEVICE, "no HIP device: the band split has no CPU fallback");
  if (device < 0 || device >= count) return qmf_fail(ASP_ERR_PARAM, "device ordinal out of range");
  QMF_TRY(hipSetDevice(device));
  AspQmfBatch* b = new AspQmfBatch();
  b->C = num_channels;
  b->device = device;
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&b->state, (size_t)num_channels * 24 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->s_in, (size_t)num_channels * 2 * ASP_QMF_MAX_BAND * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->s_a, (size_t)num_channels * ASP_QMF_MAX_BAND * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->s_b, (size_t)num_channels * ASP_QMF_MAX_BAND * sizeof(int16_t));
  if (e == hipSuccess) e = hipMemsetAsync(b->state, 0, (size_t)num_channels * 24 * sizeof(int32_t), b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e != hipSuccess) {
    AspQmfBatch_Free(b);
    return qmf_fail(ASP_ERR_HIP, "AspQmfBatch_Create", e);
  }
  *out = b;
  return ASP_OK;
}

int AspQmfBatch_Free(AspQmfBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return -1;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->state) (void)hipFree(b->state);
  if (b->s_in) (void)hipFree(b->s_in);
  if (b->s_a) (void)hipFree(b->s_a);
  if (b->s_b) (void)hipFree(b->s_b);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return 0;
}

int AspQmfBatch_num_channels(const AspQmfBatch* b) { return b ? b->C : 0; }

int AspQmfBatch_Reset(AspQmfBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return qmf_fail(ASP_ERR_PARAM, "null batch handle");
  QMF_TRY(hipSetDevice(b->device));
  QMF_TRY(hipMemsetAsync(b->state, 0, (size_t)b->C * 24 * sizeof(int32_t), b->stream));
  return ASP_OK;
}

int AspQmfBatch_Analysis(AspQmfBatch* b, const int16_t* in, int band_length, int16_t* low,
                         int16_t* high, int mem) {
  AspDeviceScope dev_scope_;
  if (!b || !in || !low || !high || band_length <= 0 || band_length > ASP_QMF_MAX_BAND)
    return qmf_fail(ASP_ERR_PARAM, "AspQmfBatch_Analysis: bad argument");
  QMF_TRY(hipSetDevice(b->device));
  const size_t nb = (size_t)b->C * band_length * sizeof(int16_t);
  const int16_t* din = in;
  int16_t *dl = low, *dh = high;
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(b->s_in, in, 2 * nb, hipMemcpyHostToDevice, b->stream));
    din = b->s_in;
    dl = b->s_a;
    dh = b->s_b;
  } else if (mem != ASP_MEM_DEVICE) {
    return qmf_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  QMF_TRY(aspqmf::launch_analysis(b->state, din, dl, dh, b->C, band_length, b->stream));
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(low, dl, nb, hipMemcpyDeviceToHost, b->stream));
    QMF_TRY(hipMemcpyAsync(high, dh, nb, hipMemcpyDeviceToHost, b->stream));
    QMF_TRY(hipStreamSynchronize(b->stream));
  }
  return ASP_OK;
}

int AspQmfBatch_Synthesis(AspQmfBatch* b, const int16_t* low, const int16_t* high,
                          int band_length, int16_t* out, int mem) {
  AspDeviceScope dev_scope_;
  if (!b || !out || !low || !high || band_length <= 0 || band_length > ASP_QMF_MAX_BAND)
    return qmf_fail(ASP_ERR_PARAM, "AspQmfBatch_Synthesis: bad argument");
  QMF_TRY(hipSetDevice(b->device));
  const size_t nb = (size_t)b->C * band_length * sizeof(int16_t);
  const int16_t *dl = low, *dh = high;
  int16_t* dout = out;
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(b->s_a, low, nb, hipMemcpyHostToDevice, b->stream));
    QMF_TRY(hipMemcpyAsync(b->s_b, high, nb, hipMemcpyHostToDevice, b->stream));
    dl = b->s_a;
    dh = b->s_b;
    dout = b->s_in;
  } else if (mem != ASP_MEM_DEVICE) {
    return qmf_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  QMF_TRY(aspqmf::launch_synthesis(b->state, dl, dh, dout, b->C, band_length, b->stream));
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(out, dout, 2 * nb, hipMemcpyDeviceToHost, b->stream));
    QMF_TRY(hipStreamSynchronize(b->stream));
  }
  return ASP_OK;
}

int AspQmfBatch_ExportState(AspQmfBatch* b, int channel, AspQmfState* out) {
  AspDeviceScope dev_scope_;
  if (!b || !out || channel < 0 || channel >= b->C) return qmf_fail(ASP_ERR_PARAM, "ExportState: bad argument");
  QMF_TRY(hipSetDevice(b->device));
  QMF_TRY(hipStreamSynchronize(b->stream));
  QMF_TRY(hipMemcpy(out, b->state + (size_t)channel * 24, sizeof *out, hipMemcpyDeviceToHost));
  return ASP_OK;
}

int AspQmfBatch_ImportState(AspQmfBatch* b, int channel, const AspQmfState* in) {
  AspDeviceScope dev_scope_;
  if (!b || !in || channel < 0 || channel >= b->C) return qmf_fail(ASP_ERR_PARAM, "ImportState: bad argument");
  QMF_TRY(hipSetDevice(b->device));
  QMF_TRY(hipStreamSynchronize(b->stream));
  QMF_TRY(hipMemcpy(b->state + (size_t)channel * 24, in, sizeof *in, hipMemcpyHostToDevice));
  return ASP_OK;
}

int AspQmfBatch_Synchronize(AspQmfBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return qmf_fail(ASP_ERR_PARAM, "null batch handle");
  QMF_TRY(hipSetDevice(b->device));
  QMF_TRY(hipStreamSynchronize(b->stream));
  return ASP_OK;
}

// ------------------------------------------------------------------ layer 1
// The reference's functions: one channel, caller-owned states.  Errors (no device) abort loudly:
// the reference functions return void.
static AspQmfBatch* one_channel() {
  static thread_local AspQmfBatch* b = nullptr;
  if (!b && AspQmfBatch_Create(&b, 1, 0) != ASP_OK) {
    fprintf(stderr, "asp_split: WebRtcSpl_*QMF needs a HIP device (no CPU fallback)\n");
    abort();
  }
  return b;
}

void WebRtcSpl_AnalysisQMF(const int16_t* in_data, int in_data_length, int16_t* low_band,
                           int16_t* high_band, int32_t* filter_state1, int32_t* filter_state2) {
  AspDeviceScope dev_scope_;
  AspQmfBatch* b = one_channel();
  AspQmfState st;
  memset(&st, 0, sizeof st);
  memcpy(st.analysis_state1, filter_state1, sizeof st.analysis_state1);
  memcpy(st.analysis_state2, filter_state2, sizeof st.analysis_state2);
  if (AspQmfBatch_ImportState(b, 0, &st) != ASP_OK ||
      AspQmfBatch_Analysis(b, in_data, in_data_length / 2, low_band, high_band, ASP_MEM_HOST) != ASP_OK ||
      AspQmfBatch_ExportState(b, 0, &st) != ASP_OK)
    abort();
  memcpy(filter_state1, st.analysis_state1, sizeof st.analysis_state1);
  memcpy(filter_state2, st.analysis_state2, sizeof st.analysis_state2);
}

void WebRtcSpl_SynthesisQMF(const int16_t* low_band, const int16_t* high_band, int band_length,
                            int16_t* out_data, int32_t* filter_state1, int32_t* filter_state2) {
  AspDeviceScope dev_scope_;
  AspQmfBatch* b = one_channel();
  AspQmfState st;
  memset(&st, 0, sizeof st);
  memcpy(st.synthesis_state1, filter_state1, sizeof st.synthesis_state1);
  memcpy(st.synthesis_state2, filter_state2, sizeof st.synthesis_state2);
  if (AspQmfBatch_ImportState(b, 0, &st) != ASP_OK ||
      AspQmfBatch_Synthesis(b, low_band, high_band, band_length, out_data, ASP_MEM_HOST) != ASP_OK ||
      AspQmfBatch_ExportState(b, 0, &st) != ASP_OK)
    abort();
  memcpy(filter_state1, st.synthesis_state1, sizeof st.synthesis_state1);
  memcpy(filter_state2, st.synthesis_state2, sizeof st.synthesis_state2);
}

// ------------------------------------------------------------------ SplittingFilter
}  // extern "C"

struct AspSplitBatch {
  int C = 0, nb = 0, device = 0;
  hipStream_t stream = nullptr;
  int32_t* st = nullptr;       // [3][C][24]: two_bands_states_, band1_states_, band2_states_
  int16_t *buf640 = nullptr, *low320 = nullptr, *high320 = nullptr, *zeros160 = nullptr;
  int16_t *s_in = nullptr, *s_bands = nullptr;   // staging for host-memory callers
  AspSincBatch *up = nullptr, *down = nullptr;   // analysis / synthesis resamplers (48 <-> 64 kHz)
};

extern "C" {

int AspSplitBatch_Create(AspSplitBatch** out, int num_channels, int num_bands, int device) {
  AspDeviceScope dev_scope_;
  if (!out || num_channels <= 0 || (num_bands != 2 && num_bands != 3))
    return qmf_fail(ASP_ERR_PARAM, "AspSplitBatch_Create: bad argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return qmf_fail(ASP_ERR_NO_DEVICE, "no HIP device: the band split has no CPU fallback");
  if (device < 0 || device >= count) return qmf_fail(ASP_ERR_PARAM, "device ordinal out of range");
  QMF_TRY(hipSetDevice(device));
  AspSplitBatch* b = new AspSplitBatch();
  b->C = num_channels;
  b->nb = num_bands;
  b->device = device;
  const size_t c = (size_t)num_channels;
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&b->st, 3 * c * 24 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->buf640, c * 640 * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->low320, c * 320 * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->high320, c * 320 * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->zeros160, c * 160 * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->s_in, c * 160 * num_bands * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc((void**)&b->s_bands, c * 160 * num_bands * sizeof(int16_t));
  if (e == hipSuccess) e = hipMemsetAsync(b->st, 0, 3 * c * 24 * sizeof(int32_t), b->stream);
  if (e == hipSuccess) e = hipMemsetAsync(b->zeros160, 0, c * 160 * sizeof(int16_t), b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  int rc = e == hipSuccess ? ASP_OK : qmf_fail(ASP_ERR_HIP, "AspSplitBatch_Create", e);
  if (rc == ASP_OK && num_bands == 3) {
    rc = AspSincBatch_Create(&b->up, num_channels, 480, 640, device);
    if (rc == ASP_OK) rc = AspSincBatch_Create(&b->down, num_channels, 640, 480, device);
    if (rc == ASP_OK) rc = AspSincBatch_SetStream(b->up, b->stream);
    if (rc == ASP_OK) rc = AspSincBatch_SetStream(b->down, b->stream);
  }
  if (rc != ASP_OK) {
    AspSplitBatch_Free(b);
    return rc;
  }
  *out = b;
  return ASP_OK;
}

int AspSplitBatch_Free(AspSplitBatch* b) {
  AspDeviceScope dev_scope_;
  if (!b) return -1;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->up) AspSincBatch_Free(b->up);
  if (b->down) AspSincBatch_Free(b->down);
  if (b->st) (void)hipFree(b->st);
  if (b->buf640) (void)hipFree(b->buf640);
  if (b->low320) (void)hipFree(b->low320);
  if (b->high320) (void)hipFree(b->high320);
  if (b->zeros160) (void)hipFree(b->zeros160);
  if (b->s_in) (void)hipFree(b->s_in);
  if (b->s_bands) (void)hipFree(b->s_bands);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return 0;
}

int AspSplitBatch_Analysis(AspSplitBatch* b, const int16_t* in, int16_t* bands, int mem) {
  AspDeviceScope dev_scope_;
  if (!b || !in || !bands) return qmf_fail(ASP_ERR_PARAM, "AspSplitBatch_Analysis: bad argument");
  QMF_TRY(hipSetDevice(b->device));
  const size_t c = (size_t)b->C, total = c * 160 * b->nb * sizeof(int16_t);
  const int16_t* din = in;
  int16_t* dbands = bands;
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(b->s_in, in, total, hipMemcpyHostToDevice, b->stream));
    din = b->s_in;
    dbands = b->s_bands;
  } else if (mem != ASP_MEM_DEVICE) {
    return qmf_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  int32_t *stA = b->st, *stB = b->st + c * 24, *stC = b->st + 2 * c * 24;
  if (b->nb == 2) {  // TwoBandsAnalysis, splitting_filter.cc:63-75
    QMF_TRY(aspqmf::launch_analysis(stA, din, dbands, dbands + c * 160, b->C, 160, b->stream));
  } else {  // ThreeBandsAnalysis, splitting_filter.cc:96-131
    const int rc = AspSincBatch_Resample(b->up, din, b->buf640, ASP_MEM_DEVICE);
    if (rc != ASP_OK) return rc;
    QMF_TRY(aspqmf::launch_analysis(stA, b->buf640, b->low320, b->high320, b->C, 320, b->stream));
    // the two second-stage banks as one launch; the lower output of the upper half is the empty
    // 24-32 kHz band: dropped
    QMF_TRY(aspqmf::launch_analysis_pair(stB, b->low320, dbands, dbands + c * 160, stC, b->high320, nullptr,
                                         dbands + 2 * c * 160, b->low320, b->C, 160, b->stream));
  }
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(bands, dbands, total, hipMemcpyDeviceToHost, b->stream));
    QMF_TRY(hipStreamSynchronize(b->stream));
  }
  return ASP_OK;
}

int AspSplitBatch_Synthesis(AspSplitBatch* b, const int16_t* bands, int16_t* out, int mem) {
  AspDeviceScope dev_scope_;
  if (!b || !out || !bands) return qmf_fail(ASP_ERR_PARAM, "AspSplitBatch_Synthesis: bad argument");
  QMF_TRY(hipSetDevice(b->device));
  const size_t c = (size_t)b->C, total = c * 160 * b->nb * sizeof(int16_t);
  const int16_t* dbands = bands;
  int16_t* dout = out;
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(b->s_bands, bands, total, hipMemcpyHostToDevice, b->stream));
    dbands = b->s_bands;
    dout = b->s_in;
  } else if (mem != ASP_MEM_DEVICE) {
    return qmf_fail(ASP_ERR_PARAM, "mem must be ASP_MEM_HOST or ASP_MEM_DEVICE");
  }
  int32_t *stA = b->st, *stB = b->st + c * 24, *stC = b->st + 2 * c * 24;
  if (b->nb == 2) {  // TwoBandsSynthesis, splitting_filter.cc:77-88
    QMF_TRY(aspqmf::launch_synthesis(stA, dbands, dbands + c * 160, dout, b->C, 160, b->stream));
  } else {  // ThreeBandsSynthesis, splitting_filter.cc:137-169 (the uppermost band is empty)
    QMF_TRY(aspqmf::launch_synthesis_pair(stB, dbands, dbands + c * 160, b->low320, stC, b->zeros160,
                                          dbands + 2 * c * 160, b->high320, b->C, 160, b->stream));
    QMF_TRY(aspqmf::launch_synthesis(stA, b->low320, b->high320, b->buf640, b->C, 320, b->stream));
    const int rc = AspSincBatch_Resample(b->down, b->buf640, dout, ASP_MEM_DEVICE);
    if (rc != ASP_OK) return rc;
  }
  if (mem == ASP_MEM_HOST) {
    QMF_TRY(hipMemcpyAsync(out, dout, total, hipMemcpyDeviceToHost, b->stream));
    QMF_TRY(hipStreamSynchronize(b->stream));
  }
  return ASP_OK;
}

}  // extern "C"
