// sinc_kernels.hip -- gfx950 kernel of the batched push sinc resampler.  Replaces, for many
// independent channels per launch, the data movement and convolutions of SincResampler::Resample
// (sinc_resampler.cc:252-312) with the SSE summation order of Convolve_SSE
// (sinc_resampler_sse.cc:19-57).  One workgroup per channel; the channel's input buffer
// (request + 32 floats) lives in LDS for the call.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sinc_layout.h"

using namespace aspsinc;

namespace {

__device__ __forceinline__ int16_t float_s16_to_s16(float v) {  // audio_util.h:41-49
  const float kMaxRound = 32767 - 0.5f, kMinRound = -32768 + 0.5f;
  if (v > 0) return v >= kMaxRound ? (int16_t)32767 : (int16_t)(v + 0.5f);
  return v <= kMinRound ? (int16_t)-32768 : (int16_t)(v - 0.5f);
}

__global__ __launch_bounds__(256) void sinc_resample_kernel(float* __restrict__ state,
                                                            const float* __restrict__ kernel_table,
                                                            const OutDesc* __restrict__ desc,
                                                            const int16_t* __restrict__ in,
                                                            int16_t* __restrict__ out,
                                                            int buf_len, int src_frames,
                                                            int dst_frames, SincPlan plan) {
  extern __shared__ float buf[];  // [buf_len]
  const int ch = blockIdx.x, tid = threadIdx.x;
  float* st = state + (size_t)ch * buf_len;
  for (int i = tid; i < buf_len; i += 256) buf[i] = st[i];
  __syncthreads();
  for (int s = 0; s < plan.nseg; ++s) {
    const SincSeg sg = plan.seg[s];
    if (sg.load != 0) {
      // memcpy(r1, r3, kKernelSize) then Run(request, r0)  (sinc_resampler.cc:303-311)
      float keep = 0.f;
      if (sg.shift && tid < kKernelSize) keep = buf[sg.r3 + tid];
      __syncthreads();
      if (sg.shift && tid < kKernelSize) buf[tid] = keep;
      const int16_t* src = in + (size_t)ch * src_frames;
      for (int i = tid; i < src_frames; i += 256) buf[sg.r0 + i] = sg.load == 2 ? (float)src[i] : 0.f;
      __syncthreads();
    }
    for (int m = sg.out_begin + tid; m < sg.out_end; m += 256) {
      const OutDesc d = desc[m];
      const float* x = buf + d.source_idx;
      const float* k1 = kernel_table + d.offset_idx * kKernelSize;
      const float* k2 = k1 + kKernelSize;
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < kKernelSize; i += 4)
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          s1[l] = s1[l] + x[i + l] * k1[i + l];
          s2[l] = s2[l] + x[i + l] * k2[i + l];
        }
      float t[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) t[l] = s1[l] * d.f1 + s2[l] * d.f2;
      const float r = (t[2] + t[0]) + (t[3] + t[1]);
      if (d.dest >= 0) out[(size_t)ch * dst_frames + d.dest] = float_s16_to_s16(r);
    }
    __syncthreads();
  }
  for (int i = tid; i < buf_len; i += 256) st[i] = buf[i];
}

}  // namespace

namespace aspsinc {

hipError_t launch_sinc(float* state, const float* kernel_table, const OutDesc* desc,
                       const int16_t* in, int16_t* out, int num_channels, int buf_len,
                       int src_frames, int dst_frames, const SincPlan& plan, hipStream_t s) {
  hipLaunchKernelGGL(sinc_resample_kernel, dim3(num_channels), dim3(256), (size_t)buf_len * sizeof(float), s,
                     state, kernel_table, desc, in, out, buf_len, src_frames, dst_frames, plan);
  return hipGetLastError();
}

}  // namespace aspsinc
