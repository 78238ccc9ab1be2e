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


// ---------------------------------------------------------------- four samples per trip, one or two jobs per launch
// The loops above wait for a 4-byte load (and, through the shared counter, for the 2-byte store before
// it) in every trip: ~260 ns per sample against ~35 instructions of arithmetic (55 us for 160 samples,
// 128 waves on the whole chip).  Here a trip is four samples: one 16-byte (analysis) or two 8-byte
// (synthesis) loads per lane, requested one trip ahead, and one 8- / 16-byte store; the arithmetic is
// the same sequence.  blockIdx.y picks one of up to two independent jobs (the two second-stage filters
// of the three-band split run as one launch).
struct QmfJob {
  int32_t* state;      // [channel][24] block of this filter bank
  const int16_t* a;    // analysis: interleaved input; synthesis: low band
  const int16_t* b;    // synthesis: high band
  int16_t* o1;         // analysis: low band (may be null: dropped); synthesis: output
  int16_t* o2;         // analysis: high band
};
struct QmfJobs {
  QmfJob j[2];
};

__device__ __forceinline__ int32_t sext16(uint32_t v) { return (int32_t)(int16_t)(v & 0xffffu); }

__global__ __launch_bounds__(64) void qmf_analysis4_kernel(QmfJobs jobs, int num_channels, int groups) {
  const QmfJob& J = jobs.j[blockIdx.y];
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int ch = t >> 1, r = t & 1;
  const bool active = ch < num_channels;
  const int c = active ? ch : 0;
  int32_t* st = J.state + (size_t)c * 24 + 6 * r;
  Cascade cas;
  load_cascade(cas, st, r == 0);
  const uint4* src = reinterpret_cast<const uint4*>(J.a + (size_t)c * 8 * groups);
  int16_t* band = r == 0 ? J.o1 : J.o2;
  uint2* dst = reinterpret_cast<uint2*>(band + (size_t)c * 4 * groups);
  const bool stores = active && band != nullptr;
  uint4 cur = src[0];
  uint2 done = {};  // the trip before's outputs: stored AFTER the next load is requested, so that waiting for
                    // the load (in-order counter) does not wait for the store's acknowledgement as well
  for (int g = 0; g < groups; ++g) {
    uint4 nxt = cur;
    if (g + 1 < groups) nxt = src[g + 1];
    if (stores && g > 0) dst[g - 1] = done;
    const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
    int32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int32_t s = r == 0 ? (int32_t)w[k] >> 16 : sext16(w[k]);
      const int32_t f = cas.step((int32_t)((uint32_t)s << 10));
      const int32_t other = lane_swap1(f);
      const int32_t f1 = r == 0 ? f : other, f2 = r == 0 ? other : f;
      v[k] = sat16(r == 0 ? (f1 + f2 + 1024) >> 11 : (f1 - f2 + 1024) >> 11);
    }
    done = uint2{((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16), ((uint32_t)v[2] & 0xffffu) | ((uint32_t)v[3] << 16)};
    cur = nxt;
  }
  if (stores) dst[groups - 1] = done;
  if (active) store_cascade(cas, st);
}

__global__ __launch_bounds__(64) void qmf_synthesis4_kernel(QmfJobs jobs, int num_channels, int groups) {
  const QmfJob& J = jobs.j[blockIdx.y];
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int ch = t >> 1, r = t & 1;
  const bool active = ch < num_channels;
  const int c = active ? ch : 0;
  int32_t* st = J.state + (size_t)c * 24 + 12 + 6 * r;
  Cascade cas;
  load_cascade(cas, st, r == 1);
  const uint2* lo = reinterpret_cast<const uint2*>(J.a + (size_t)c * 4 * groups);
  const uint2* hi = reinterpret_cast<const uint2*>(J.b + (size_t)c * 4 * groups);
  uint4* dst = reinterpret_cast<uint4*>(J.o1 + (size_t)c * 8 * groups);
  uint2 cl = lo[0], chh = hi[0];
  uint4 done = {};
  const bool stores = active && r == 0;
  for (int g = 0; g < groups; ++g) {
    uint2 nl = cl, nh = chh;
    if (g + 1 < groups) {
      nl = lo[g + 1];
      nh = hi[g + 1];
    }
    if (stores && g > 0) dst[g - 1] = done;
    const int32_t l[4] = {sext16(cl.x), (int32_t)cl.x >> 16, sext16(cl.y), (int32_t)cl.y >> 16};
    const int32_t h[4] = {sext16(chh.x), (int32_t)chh.x >> 16, sext16(chh.y), (int32_t)chh.y >> 16};
    uint32_t word[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int32_t tmp = r == 0 ? l[k] + h[k] : l[k] - h[k];
      const int32_t f = cas.step((int32_t)((uint32_t)tmp << 10));
      const int32_t mine = sat16((f + 512) >> 10);
      const int32_t other = lane_swap1(mine);
      word[k] = ((uint32_t)mine << 16) | ((uint32_t)other & 0xffffu);  // the r = 0 lane's is the output pair
    }
    done = uint4{word[0], word[1], word[2], word[3]};
    cl = nl;
    chh = nh;
  }
  if (stores) dst[groups - 1] = done;
  if (active) store_cascade(cas, st);
}

}  // namespace

namespace aspqmf {

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool analysis4_ok(const int16_t* in, const int16_t* low, const int16_t* high, int band_length) {
  return band_length % 4 == 0 && aligned16(in) && aligned16(low) && aligned16(high);
}
static bool synthesis4_ok(const int16_t* low, const int16_t* high, const int16_t* out, int band_length) {
  return band_length % 4 == 0 && aligned16(low) && aligned16(high) && aligned16(out);
}

hipError_t launch_analysis(int32_t* state, const int16_t* in, int16_t* low, int16_t* high,
                           int num_channels, int band_length, hipStream_t s) {
  if (analysis4_ok(in, low, high, band_length) && low != nullptr) {
    QmfJobs jobs = {};
    jobs.j[0] = QmfJob{state, in, nullptr, low, high};
    hipLaunchKernelGGL(qmf_analysis4_kernel, dim3((2 * num_channels + 63) / 64, 1), dim3(64), 0, s, jobs, num_channels,
                       band_length / 4);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(qmf_analysis_kernel, dim3((2 * num_channels + 63) / 64), dim3(64), 0, s, state, in,
                     low, high, num_channels, band_length);
  return hipGetLastError();
}

hipError_t launch_synthesis(int32_t* state, const int16_t* low, const int16_t* high, int16_t* out,
                            int num_channels, int band_length, hipStream_t s) {
  if (synthesis4_ok(low, high, out, band_length)) {
    QmfJobs jobs = {};
    jobs.j[0] = QmfJob{state, low, high, out, nullptr};
    hipLaunchKernelGGL(qmf_synthesis4_kernel, dim3((2 * num_channels + 63) / 64, 1), dim3(64), 0, s, jobs,
                       num_channels, band_length / 4);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(qmf_synthesis_kernel, dim3((2 * num_channels + 63) / 64), dim3(64), 0, s, state,
                     low, high, out, num_channels, band_length);
  return hipGetLastError();
}

// Two independent filter banks of one band length in one launch (the second stage of the three-band
// split / merge).  `low1` may be null: that output is dropped (the empty 24-32 kHz band).
hipError_t launch_analysis_pair(int32_t* state0, const int16_t* in0, int16_t* low0, int16_t* high0,
                                int32_t* state1, const int16_t* in1, int16_t* low1, int16_t* high1,
                                int16_t* drop_scratch, int num_channels, int band_length, hipStream_t s) {
  if (!(analysis4_ok(in0, low0, high0, band_length) && analysis4_ok(in1, low1, high1, band_length))) {
    // unaligned caller buffers: one bank after the other (`drop_scratch` may alias in0: bank 0 is done with it)
    const hipError_t e = launch_analysis(state0, in0, low0, high0, num_channels, band_length, s);
    if (e != hipSuccess) return e;
    return launch_analysis(state1, in1, low1 ? low1 : drop_scratch, high1, num_channels, band_length, s);
  }
  QmfJobs jobs = {};
  jobs.j[0] = QmfJob{state0, in0, nullptr, low0, high0};
  jobs.j[1] = QmfJob{state1, in1, nullptr, low1, high1};
  hipLaunchKernelGGL(qmf_analysis4_kernel, dim3((2 * num_channels + 63) / 64, 2), dim3(64), 0, s, jobs, num_channels,
                     band_length / 4);
  return hipGetLastError();
}

hipError_t launch_synthesis_pair(int32_t* state0, const int16_t* low0, const int16_t* high0, int16_t* out0,
                                 int32_t* state1, const int16_t* low1, const int16_t* high1, int16_t* out1,
                                 int num_channels, int band_length, hipStream_t s) {
  if (!(synthesis4_ok(low0, high0, out0, band_length) && synthesis4_ok(low1, high1, out1, band_length))) {
    const hipError_t e = launch_synthesis(state0, low0, high0, out0, num_channels, band_length, s);
    if (e != hipSuccess) return e;
    return launch_synthesis(state1, low1, high1, out1, num_channels, band_length, s);
  }
  QmfJobs jobs = {};
  jobs.j[0] = QmfJob{state0, low0, high0, out0, nullptr};
  jobs.j[1] = QmfJob{state1, low1, high1, out1, nullptr};
  hipLaunchKernelGGL(qmf_synthesis4_kernel, dim3((2 * num_channels + 63) / 64, 2), dim3(64), 0, s, jobs,
                     num_channels, band_length / 4);
  return hipGetLastError();
}

}  // namespace aspqmf
