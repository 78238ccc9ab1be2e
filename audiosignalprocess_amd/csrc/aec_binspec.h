// aec_binspec.h -- BinarySpectrumFloat of the delay estimator (utility/delay_estimator_wrapper.c:43-48, 96-124),
// shared by the estimator kernel (aec_delay_kernels.hip) and the hand-off build of the process kernel
// (aec_kernels.hip), which forms the two binary spectra of a block where |X|^2 and |D|^2 are in registers.
#pragma once
#include <hip/hip_runtime.h>

namespace aspaec {

constexpr int kBinSpecBandFirst = 12, kBinSpecBandLast = 43;  // delay_estimator_wrapper.c:20-23

// lane = band; `thr` = the band's mean spectrum (MeanEstimatorFloat), `initialized` wave-uniform.  Every lane of the wave active.
__device__ __forceinline__ unsigned binary_spectrum(float spec, float& thr, int& initialized, int lane) {
  const bool band = lane >= kBinSpecBandFirst && lane <= kBinSpecBandLast;
  const float kScale = 1 / 64.0;
  if (!initialized) {
    const bool pos = band && spec > 0.0f;
    if (pos) thr = spec / 2;
    if (__ballot(pos) != 0) initialized = 1;
  }
  bool bit = false;
  if (band) {
    thr += (spec - thr) * kScale;  // MeanEstimatorFloat, :43-48
    bit = spec > thr;
  }
  return (unsigned)((__ballot(bit) >> kBinSpecBandFirst) & 0xffffffffull);
}

}  // namespace aspaec
