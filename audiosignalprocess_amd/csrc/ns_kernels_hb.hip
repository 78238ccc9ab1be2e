// ns_kernels_hb.hip -- the high-band branch of WebRtcNs_ProcessCore for 32 / 48 kHz input
// (ns_core.c:1227-1235, 1252-1261, 1362-1414): the band split hands the suppressor the 0-8 kHz band
// plus one or two high bands; the low band runs through the frame kernels unchanged, the high
// bands get one time-domain gain derived from the low band's speech probability and gain filter.
//
// Two small kernels bracket the low-band step, so the tuned frame kernels are untouched:
//   ns_hb_live_kernel   (before): energy1 == 0 of the windowed low-band analysis buffer decides
//                        between the early-exit copy and the gained path (ns_core.c:1237-1264); a
//                        sum of squares is zero iff every square is, so no summation order matters;
//   ns_hb_apply_kernel  (after):  speechProb of bins 96..127 re-derived from the stream's own state
//                        rows exactly as SpeechNoiseProb left it (ns_core.c:742-748), the three
//                        32-term / 129-term sums accumulated by one lane each in the reference's
//                        order, tanh gain map, flooring, delayed high-band samples scaled.
// One wave64 per stream.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "ns_device.h"
#include "ns_layout.h"

using namespace aspns;
using namespace aspns_dev;

namespace {

__global__ __launch_bounds__(256) void ns_hb_live_kernel(const float* __restrict__ state,
                                                         const NsTables* __restrict__ T,
                                                         const float* __restrict__ in_low,
                                                         int32_t* __restrict__ live,
                                                         int num_streams, int hist_off) {
  const int lane = threadIdx.x & 63;
  const int stream = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (stream >= num_streams) return;
  const float* st = state + (size_t)stream * kStreamDwords;
  bool nz = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = lane + 64 * k;
    const float x = j < kCarry ? st[hist_off + j] : in_low[(size_t)stream * kBlockL + (j - kCarry)];
    const float p = T->window[j] * x;  // Windowing, ns_core.c:969-978
    nz = nz || (p * p != 0.0f);        // a term of Energy(), ns_core.c:951-960
  }
  const unsigned long long any = __ballot(nz);
  if (lane == 0) live[stream] = any != 0ull;
}

// hb_tail [stream][2][96]: dataBufHB[b][160..255]; in_high / out_high [num_high][num_streams][160]
__global__ __launch_bounds__(256) void ns_hb_apply_kernel(const float* __restrict__ state,
                                                          float* __restrict__ hb_tail,
                                                          const int32_t* __restrict__ live_flags,
                                                          const NsTables* __restrict__ T,
                                                          const float* __restrict__ in_high,
                                                          float* __restrict__ out_high,
                                                          int num_streams, int num_high, int paired) {
  // per wave: speechProb[32] | smooth[32] | magnPrevAnalyze[132] | magnPrevProcess[132] | 4 sums
  constexpr int kSums = 64 + 2 * 132;
  __shared__ float lds[4][kSums + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stream = blockIdx.x * 4 + wave;
  if (stream >= num_streams) return;
  const float* st = state + (size_t)stream * kStreamDwords;
  const float* vec = st + kOffVec;
  float* l = lds[wave];
  const bool live = live_flags[stream] != 0;
  float gain = 1.0f;
  if (live) {
    const float prior = st[kOffScalars + S_PRIORSPEECHPROB];
    const float denoiseBound = st[kOffScalars + S_DENOISEBOUND];
    // speechProb[i], smooth[i] for i = 96 .. 127 (magnLen - deltaBweHB - 1 .. magnLen - 2)
    if (lane < 32) {
      const int i = 96 + lane;
      const float gainPrior = (1.f - prior) / (prior + 0.0001f);        // ns_core.c:743
      float invLrt = exp_f32_via_f64(-vec[V_LOGLRT * kVecStride + row_pos(i)], T->exp2_64);
      invLrt = gainPrior * invLrt;
      l[lane] = 1.f / (1.f + invLrt);                                   // speechProb[i]
      l[32 + lane] = vec[V_SMOOTH * kVecStride + row_pos(i)];
    }
    if (!paired) {
      for (int i = lane; i < kBins; i += 64) {
        l[64 + i] = st[row_dword(V_MAGNPREV_A, i)];
        l[64 + 132 + i] = st[row_dword(V_MAGNPREV_P, i)];
      }
    }
    wave_lds_fence();
    if (lane < 4) {  // the reference's sequential sums, one lane each
      float acc = 0.f;
      if (lane < 2) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc += l[32 * lane + j];
      } else if (!paired) {
        const float* src = l + 64 + 132 * (lane - 2);
#pragma unroll 8
        for (int j = 0; j < kBins; ++j) acc += src[j];
      }
      l[kSums + lane] = acc;
    }
    wave_lds_fence();
    float avgProbSpeechHB = l[kSums + 0] / 32.0f;                // ns_core.c:1367-1371
    if (!paired) {
      // paired state: magnPrevProcess is magnPrevAnalyze, the ratio is exactly 1
      avgProbSpeechHB *= l[kSums + 3] / l[kSums + 2];     // :1375-1381
    }
    const float avgFilterGainHB = l[kSums + 1] / 32.0f;          // :1384-1388
    const float tmp = 2.f * avgProbSpeechHB - 1.f;
    const float gainModHB = 0.5f * (1.f + tanh_f32_via_f64(1.0f * tmp, T->exp2_64));  // :1391
    gain = 0.5f * gainModHB + 0.5f * avgFilterGainHB;
    if (avgProbSpeechHB >= 0.5f) gain = 0.25f * gainModHB + 0.75f * avgFilterGainHB;
    gain = gain * 1.0f;  // decayBweHB
    if (gain < denoiseBound) gain = denoiseBound;
    if (gain > 1.f) gain = 1.f;
  }
  for (int b = 0; b < num_high; ++b) {
    float* tail = hb_tail + ((size_t)stream * 2 + b) * kCarry;
    const float* x = in_high + ((size_t)b * num_streams + stream) * kBlockL;
    float* y = out_high + ((size_t)b * num_streams + stream) * kBlockL;
    // dataBufHB after UpdateBuffer = [tail (96) | x (160)]; the output reads its first 160
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
      v[k] = 0.f;
      if (j < kBlockL) v[k] = j < kCarry ? tail[j] : x[j - kCarry];
    }
    const float t0 = x[64 + lane];
    const float t1 = lane < 32 ? x[128 + lane] : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // in_high may alias out_high; tail RAW
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
      if (j < kBlockL) {
        const float o = live ? gain * v[k] : v[k];                       // :1409-1411 / 1255-1259
        y[j] = o > 32767.f ? 32767.f : (o < -32768.f ? -32768.f : o);
      }
    }
    tail[lane] = t0;  // new tail = x[64 .. 159]
    if (lane < 32) tail[64 + lane] = t1;
  }
}

}  // namespace

namespace aspns {

hipError_t launch_ns_hb_live(const float* state, const NsTables* T, const float* in_low,
                             int32_t* live, int num_streams, int hist_off, hipStream_t s) {
  hipLaunchKernelGGL(ns_hb_live_kernel, dim3((num_streams + 3) / 4), dim3(256), 0, s, state, T, in_low,
                     live, num_streams, hist_off);
  return hipGetLastError();
}

hipError_t launch_ns_hb_apply(const float* state, float* hb_tail, const int32_t* live,
                              const NsTables* T, const float* in_high, float* out_high,
                              int num_streams, int num_high, int paired, hipStream_t s) {
  hipLaunchKernelGGL(ns_hb_apply_kernel, dim3((num_streams + 3) / 4), dim3(256), 0, s, state, hb_tail,
                     live, T, in_high, out_high, num_streams, num_high, paired);
  return hipGetLastError();
}

}  // namespace aspns
