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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The kernel table [33 offsets][32 taps] staged in LDS per workgroup (4.1 KB): a lane reads the two
// rows of its output as eight 16-byte chunks each.  Lanes of a wave sit on different rows (offset
// 0 / 8 / 16 / 24 for 48 -> 64 kHz), which all start at bank 0: chunk c of row r is kept at chunk
// c ^ (r / 8 mod 8), so those rows' equal chunks fall into different banks.
constexpr int kTabFloats = (kKernelOffsetCount + 1) * kKernelSize;
__device__ __forceinline__ int tab_swz(int row) { return (row >> 3) & 7; }

__global__ __launch_bounds__(256) void sinc_resample_kernel(float* __restrict__ state,
                                                            const float* __restrict__ kernel_table,
                                                            const OutDesc* __restrict__ desc,
                                                            const int16_t* __restrict__ in,
                                                            int16_t* __restrict__ out,
                                                            int buf_len, int src_frames,
                                                            int dst_frames, SincPlan plan) {
  extern __shared__ __align__(16) float smem[];  // [kTabFloats] table, [buf_len] the channel's buffer
  float4* ktab = reinterpret_cast<float4*>(smem);
  float* buf = smem + kTabFloats;
  const int ch = blockIdx.x, tid = threadIdx.x;
  float* st = state + (size_t)ch * buf_len;
  for (int i = tid; i < kTabFloats / 4; i += 256) {
    const int row = i >> 3, c = i & 7;
    ktab[row * 8 + (c ^ tab_swz(row))] = reinterpret_cast<const float4*>(kernel_table)[i];
  }
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
      const float4* k1 = ktab + d.offset_idx * 8;
      const float4* k2 = k1 + 8;
      const int z1 = tab_swz(d.offset_idx), z2 = tab_swz(d.offset_idx + 1);
      // Convolve_SSE's four partial sums per kernel, as two packed pairs each: the multiply and the add
      // of a tap stay separate IEEE operations (no contraction)
      f32x2 s1a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2a = {0.f, 0.f}, s2b = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < kKernelSize / 4; ++c) {
        const float4 a = k1[c ^ z1], b = k2[c ^ z2];
        const f32x2 xa = {x[4 * c], x[4 * c + 1]}, xb = {x[4 * c + 2], x[4 * c + 3]};
        s1a = s1a + xa * f32x2{a.x, a.y};
        s1b = s1b + xb * f32x2{a.z, a.w};
        s2a = s2a + xa * f32x2{b.x, b.y};
        s2b = s2b + xb * f32x2{b.z, b.w};
      }
      const f32x2 ta = s1a * d.f1 + s2a * d.f2, tb = s1b * d.f1 + s2b * d.f2;  // t[0], t[1]; t[2], t[3]
      const float r = (tb.x + ta.x) + (tb.y + ta.y);
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
  hipLaunchKernelGGL(sinc_resample_kernel, dim3(num_channels), dim3(256), (size_t)(kTabFloats + buf_len) * sizeof(float), s,
                     state, kernel_table, desc, in, out, buf_len, src_frames, dst_frames, plan);
  return hipGetLastError();
}

}  // namespace aspsinc
