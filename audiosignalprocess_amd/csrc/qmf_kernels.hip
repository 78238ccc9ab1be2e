// qmf_kernels.hip -- gfx950 kernels of the batched two-band QMF split / merge.  Replaces, for many
// independent channels per launch, WebRtcSpl_AnalysisQMF / WebRtcSpl_SynthesisQMF / AllPassQMF
// (common_audio/signal_processing/splitting_filter_c.c:45-212).
//
// The filters are cascades of first-order all-pass sections in saturating fixed point: serial in
// time, so the parallelism is channels x the two polyphase branches: lane 2c + r runs branch r of
// channel c with the three sections of its cascade fused into one loop over time (a section at
// time k needs its input at k, k-1 and its own output at k-1).  The two branches of a channel sit
// on neighbouring lanes and trade their outputs with one DPP swap per sample for the sum /
// difference step.  Integer arithmetic only: bit-exact by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// WebRtcSpl_SubSatW32 (spl_inl.h): signed subtraction saturated to [INT32_MIN, INT32_MAX] -- exactly
// llvm.ssub.sat, one v_sub_i32 with the clamp bit instead of a compare-and-select ladder
__device__ __forceinline__ int32_t sub_sat(int32_t a, int32_t b) { return __builtin_elementwise_sub_sat(a, b); }

// WEBRTC_SPL_SCALEDIFF32 (signal_processing_library.h:77-79), summed in unsigned arithmetic
__device__ __forceinline__ int32_t scale_diff(uint32_t a, int32_t b, int32_t c) {
  const uint32_t hi = (uint32_t)((b >> 16) * (int32_t)a);
  const uint32_t lo = ((uint32_t)(0x0000FFFF & b) * a) >> 16;
  return (int32_t)((uint32_t)c + hi + lo);
}

__device__ __forceinline__ int32_t sat16(int32_t v) {
  return v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
}

struct Cascade {
  int32_t x1, y1a, y1b, y2a, y2b, y3;
  uint32_t a0, a1, a2;
  __device__ __forceinline__ int32_t step(int32_t x) {  // WebRtcSpl_AllPassQMF, one sample
    const int32_t y1 = scale_diff(a0, sub_sat(x, y1a), x1);
    const int32_t y2 = scale_diff(a1, sub_sat(y1, y2a), y1b);
    const int32_t y = scale_diff(a2, sub_sat(y2, y3), y2b);
    x1 = x;
    y1a = y1;
    y1b = y1;
    y2a = y2;
    y2b = y2;
    y3 = y;
    return y;
  }
};

__device__ __forceinline__ void load_cascade(Cascade& c, const int32_t* st, bool filter1) {
  c.x1 = st[0];
  c.y1a = st[1];
  c.y1b = st[2];
  c.y2a = st[3];
  c.y2b = st[4];
  c.y3 = st[5];
  // WebRtcSpl_kAllPassFilter1 / 2 (splitting_filter_c.c:25-26)
  c.a0 = filter1 ? 6418u : 21333u;
  c.a1 = filter1 ? 36982u : 49062u;
  c.a2 = filter1 ? 57261u : 63010u;
}
__device__ __forceinline__ void store_cascade(const Cascade& c, int32_t* st) {
  st[0] = c.x1;
  st[1] = c.y1a;
  st[2] = c.y1b;
  st[3] = c.y2a;
  st[4] = c.y2b;
  st[5] = c.y3;
}

__device__ __forceinline__ int32_t lane_swap1(int32_t v) {  // value of lane ^ 1
  return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
}

// state [channel][24]: analysis_state1, analysis_state2, synthesis_state1, synthesis_state2
__global__ __launch_bounds__(64) void qmf_analysis_kernel(int32_t* __restrict__ state,
                                                          const int16_t* __restrict__ in,
                                                          int16_t* __restrict__ low,
                                                          int16_t* __restrict__ high,
                                                          int num_channels, int band_length) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int ch = t >> 1, r = t & 1;
  const bool active = ch < num_channels;
  const int c = active ? ch : 0;
  // r = 0: filter1 on the odd samples with state1; r = 1: filter2 on the even samples with state2
  int32_t* st = state + (size_t)c * 24 + 6 * r;
  Cascade cas;
  load_cascade(cas, st, r == 0);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(in + (size_t)c * 2 * band_length);
  int16_t* dst = (r == 0 ? low : high) + (size_t)c * band_length;

  for (int i = 0; i < band_length; ++i) {
    const uint32_t w = words[i];  // (in[2i], in[2i+1])
    const int32_t s = r == 0 ? (int32_t)w >> 16 : (int32_t)(int16_t)(w & 0xffffu);
    const int32_t f = cas.step((int32_t)((uint32_t)s << 10));
    const int32_t other = lane_swap1(f);
    const int32_t f1 = r == 0 ? f : other, f2 = r == 0 ? other : f;
    const int32_t v = r == 0 ? (f1 + f2 + 1024) >> 11 : (f1 - f2 + 1024) >> 11;
    if (active) dst[i] = (int16_t)sat16(v);
  }
  if (active) store_cascade(cas, st);
}

__global__ __launch_bounds__(64) void qmf_synthesis_kernel(int32_t* __restrict__ state,
                                                           const int16_t* __restrict__ low,
                                                           const int16_t* __restrict__ high,
                                                           int16_t* __restrict__ out,
                                                           int num_channels, int band_length) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int ch = t >> 1, r = t & 1;
  const bool active = ch < num_channels;
  const int c = active ? ch : 0;
  // r = 0: sum channel through filter2 with state1 (odd output samples);
  // r = 1: difference channel through filter1 with state2 (even output samples)
  int32_t* st = state + (size_t)c * 24 + 12 + 6 * r;
  Cascade cas;
  load_cascade(cas, st, r == 1);
  const int16_t* lo = low + (size_t)c * band_length;
  const int16_t* hi = high + (size_t)c * band_length;
  uint32_t* words = reinterpret_cast<uint32_t*>(out + (size_t)c * 2 * band_length);

  for (int i = 0; i < band_length; ++i) {
    const int32_t l = lo[i], h = hi[i];
    const int32_t tmp = r == 0 ? l + h : l - h;
    const int32_t f = cas.step((int32_t)((uint32_t)tmp << 10));
    const int32_t mine = sat16((f + 512) >> 10);
    const int32_t other = lane_swap1(mine);
    // out[2i] = from filter2 (r = 1), out[2i + 1] = from filter1 (r = 0); the r = 0 lane stores
    if (active && r == 0) words[i] = ((uint32_t)mine << 16) | ((uint32_t)other & 0xffffu);
  }
  if (active) store_cascade(cas, st);
}

}  // namespace

namespace aspqmf {

hipError_t launch_analysis(int32_t* state, const int16_t* in, int16_t* low, int16_t* high,
                           int num_channels, int band_length, hipStream_t s) {
  hipLaunchKernelGGL(qmf_analysis_kernel, dim3((2 * num_channels + 63) / 64), dim3(64), 0, s, state, in,
                     low, high, num_channels, band_length);
  return hipGetLastError();
}

hipError_t launch_synthesis(int32_t* state, const int16_t* low, const int16_t* high, int16_t* out,
                            int num_channels, int band_length, hipStream_t s) {
  hipLaunchKernelGGL(qmf_synthesis_kernel, dim3((2 * num_channels + 63) / 64), dim3(64), 0, s, state,
                     low, high, out, num_channels, band_length);
  return hipGetLastError();
}

}  // namespace aspqmf
