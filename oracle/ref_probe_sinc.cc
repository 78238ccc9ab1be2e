// ref_probe_sinc.cc -- glue compiled INTO oracle/_ref/libsinc_ref.so next to the reference's own
// sinc resampler sources (compiled in place from /root/reference; see oracle/Makefile).  TEST
// INFRASTRUCTURE ONLY; contains no algorithm.  x86-64 builds of the reference select
// SincResampler::Convolve_SSE at compile time (sinc_resampler.cc:104-106).
#include "webrtc/common_audio/resampler/push_sinc_resampler.h"

extern "C" {
void* ref_sinc_create(int src_frames, int dst_frames) {
  return new webrtc::PushSincResampler(src_frames, dst_frames);
}
void ref_sinc_free(void* h) { delete static_cast<webrtc::PushSincResampler*>(h); }
int ref_sinc_resample_i16(void* h, const int16_t* in, int n_in, int16_t* out, int n_out) {
  return static_cast<webrtc::PushSincResampler*>(h)->Resample(in, n_in, out, n_out);
}
}
