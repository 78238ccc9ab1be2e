// sinc_layout.h -- descriptors shared by sinc_kernels.hip and sinc_api.hip.
#pragma once
#include <stdint.h>

namespace aspsinc {

constexpr int kKernelSize = 32;         // sinc_resampler.h:41
constexpr int kKernelOffsetCount = 32;  // sinc_resampler.h:50

// one output sample of SincResampler::Resample (sinc_resampler.cc:273-296)
struct OutDesc {
  int32_t source_idx;  // input_ptr = r1 + source_idx
  int32_t offset_idx;  // k1 = kernel + offset_idx * 32, k2 = k1 + 32
  float f1, f2;        // (float)(1 - kernel_interpolation_factor), (float)factor
  int32_t dest;        // index in the caller's output, -1: discarded (priming pass)
};

// a run of outputs, optionally preceded by the buffer shift and a new input block
struct SincSeg {
  int32_t load;       // 0 none, 1 zeros (first pass, push_sinc_resampler.cc:85-89), 2 the call's source
  int32_t shift;      // memcpy(r1, r3, 32) before the load
  int32_t r0, r3;
  int32_t out_begin, out_end;  // descriptor range
};

struct SincPlan {
  int32_t nseg;
  SincSeg seg[6];
};

}  // namespace aspsinc
