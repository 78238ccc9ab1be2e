// aec_delay_kernels.hip -- the AEC's delay estimation and the per-stream part of its control plane (gfx950).
//
// Replaces, per stream and bit for bit:
//   * the block-wise delay estimation of ProcessBlock (aec_core.c:1191-1203): WebRtc_AddFarSpectrumFloat +
//     WebRtc_DelayEstimatorProcessFloat = BinarySpectrumFloat (utility/delay_estimator_wrapper.c:43-48, 96-124),
//     WebRtc_AddBinaryFarSpectrum, WebRtc_ProcessBinarySpectrum with the robust validation
//     (utility/delay_estimator.c:38-146, 173-258, 356-369, 513-644), and the logging histogram;
//   * in the delay-agnostic mode (WebRtcAec_enable_reported_delay(core, 0)) the far-buffer side of
//     WebRtcAec_ProcessFrames for one 80-sample sub-frame (aec_core.c:1696-1751): the under-run stuffing,
//     SignalBasedDelayCorrection (:797-850), the read-pointer move, the estimator's soft reset
//     (delay_estimator.c:309-339, 500-511) and the far slots of the blocks to come -- and, replayed first, the
//     read-side bookkeeping of the WebRtcAec_BufferFarend calls since the last step (aec_core.c:1618-1635:
//     a full far buffer drops its oldest partition).
//
// One wave64 per stream, four per workgroup.  The process kernel leaves |X|^2 and |D|^2 of each block in a
// scratch ([stream][block][2][kRow]) -- or, in its hand-off build, the block's two binary spectra (two words) -- and
// this kernel runs after it.  Lane q owns entries q and q + 64 of the 125 / 126-entry arrays, in registers while
// the wave works: the histories move up by one entry per block with a DPP wave shift, entries named by a
// wave-uniform index (the lookahead, the compared delay, the candidate) are read with v_readlane, the best / worst
// candidates are DPP reductions; the estimator's scalars live in SGPRs and lane 0 writes them back.  No LDS, no
// barriers.  Integer and float operations are the reference's, in its order.
//
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "aec_estimator.h"

using namespace aspaec;
using namespace aspaec_est;

namespace {


__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void aec_delay_kernel(DelayBlock* __restrict__ blocks, const float* __restrict__ spectra,
                                                        int num_streams, DelayOps ops) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wave;
  if (stream >= num_streams) return;
  DelayBlock* blk = blocks + stream;
  AspAecDelayState* g = &blk->s;
  Hist H;
  Scalars sc;
  load_estimator(g, H, sc, lane);
  float thr_far = g->mean_far_spectrum[lane];
  float thr_near = g->mean_near_spectrum[lane];

  // ---- the blocks the process kernel left behind (aec_core.c:1191-1203)
  const float* sp = spectra + (size_t)stream * kSpecBlocks * kSpecDwords;
  for (int k = 0; k < ops.npending; ++k) {
    const float far_pow = sp[k * kSpecDwords + lane], near_pow = sp[k * kSpecDwords + kRow + lane];
    const int delay_estimate = estimator_block(H, sc, far_pow, near_pow, thr_far, thr_near, lane);
    if (ops.logging && delay_estimate >= 0 && lane == 0) g->delay_histogram[delay_estimate]++;
  }

  // ---- agnostic mode: the stream's far buffer for the coming sub-frame
  if (ops.control) {
    int slot0, slot1;
    control_step(blk, H, sc, ops, lane, slot0, slot1);
  }

  // ---- back to HBM
  store_estimator<true>(g, H, sc, lane);
  g->mean_far_spectrum[lane] = thr_far;
  g->mean_near_spectrum[lane] = thr_near;
}

// The estimator's share of a hand-off launch (aec_kernels.hip, aec_process_flow_kernel): `npending` blocks per stream
// whose binary spectra the process kernel left in `bits` ([stream][kFlowBitsBlocks][far, near]), in block order.
__global__ __launch_bounds__(256) void aec_delay_bits_kernel(DelayBlock* __restrict__ blocks, const unsigned* __restrict__ bits,
                                                             int num_streams, int npending, int logging) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wave;
  if (stream >= num_streams) return;
  AspAecDelayState* g = &blocks[stream].s;
  Hist H;
  Scalars sc;
  load_estimator(g, H, sc, lane);
  const unsigned* w = bits + (size_t)stream * kFlowBitsBlocks * 2;
  for (int k = 0; k < npending; ++k) {
    const unsigned bfar = __builtin_amdgcn_readfirstlane(w[2 * k]), bnear = __builtin_amdgcn_readfirstlane(w[2 * k + 1]);
    const int delay_estimate = estimator_block_bits(H, sc, bfar, bnear, lane);
    if (logging && delay_estimate >= 0 && lane == 0) g->delay_histogram[delay_estimate]++;
  }
  store_estimator<false>(g, H, sc, lane);
}

// WebRtcAec_ResampleLinear for every stream (aec_resampler.c:74-123; echo_cancellation.c:304-313, skew compensation):
// the new far frame goes behind the buffered one (one sample of look-ahead), output sample mm is the linear
// interpolation at be * mm + position -- the same positions for every stream, computed per lane with the reference's
// float expressions -- and the buffer moves up by `size`.  One wave per stream; the 320-sample buffer passes through LDS.
__global__ __launch_bounds__(256) void aec_resample_kernel(float* __restrict__ rs_buffer, const float* __restrict__ farend,
                                                           float* __restrict__ out, int num_streams, int size,
                                                           int size_out, float be, float position) {
  __shared__ float lds[4 * kResamplerBufferSize];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wave;
  if (stream >= num_streams) return;
  float* buffer = lds + wave * kResamplerBufferSize;
  float* g = rs_buffer + (size_t)stream * kResamplerBufferSize;
  for (int i = lane; i < kResamplerBufferSize; i += 64) buffer[i] = g[i];
  wave_fence();
  for (int i = lane; i < size; i += 64) buffer[kFrameLen + kResamplingDelay + i] = farend[(size_t)stream * size + i];
  wave_fence();
  const float* y = buffer + kFrameLen;  // the current frame
  for (int mm = lane; mm < size_out; mm += 64) {
    const float tnew = be * mm + position;
    const int tn = (int)tnew;
    out[(size_t)stream * size_out + mm] = y[tn] + (tnew - tn) * (y[tn + 1] - y[tn]);
  }
  for (int i = lane; i < kResamplerBufferSize - size; i += 64) g[i] = buffer[i + size];  // memmove(buffer, &buffer[size], ...)
  for (int i = kResamplerBufferSize - size + lane; i < kResamplerBufferSize; i += 64) g[i] = buffer[i];
}

}  // namespace

namespace aspaec {

hipError_t launch_aec_resample(float* rs_buffer, const float* farend, float* out, int num_streams, int size, int size_out,
                               float be, float position, hipStream_t s) {
  if (size < 0 || size > 2 * kFrameLen || size_out < 0 || size_out > kResamplerBufferSize) return hipErrorInvalidValue;
  hipLaunchKernelGGL(aec_resample_kernel, dim3((num_streams + 3) / 4), dim3(256), 0, s, rs_buffer, farend, out, num_streams,
                     size, size_out, be, position);
  return hipGetLastError();
}

hipError_t launch_aec_delay(DelayBlock* blocks, const float* spectra, int num_streams, const DelayOps& ops,
                            hipStream_t s) {
  hipLaunchKernelGGL(aec_delay_kernel, dim3((num_streams + 3) / 4), dim3(256), 0, s, blocks, spectra, num_streams, ops);
  return hipGetLastError();
}

hipError_t launch_aec_delay_bits(DelayBlock* blocks, const unsigned* bits, int num_streams, int npending, int logging,
                                 hipStream_t s) {
  hipLaunchKernelGGL(aec_delay_bits_kernel, dim3((num_streams + 3) / 4), dim3(256), 0, s, blocks, bits, num_streams, npending,
                     logging);
  return hipGetLastError();
}

}  // namespace aspaec
