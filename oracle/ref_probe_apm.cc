// ref_probe_apm.cc -- glue compiled INTO oracle/_ref/libapm_ref.so next to the reference's own
// libapm APM_NS class and everything under it (AudioBuffer, SplittingFilter, sinc resampler, the
// float noise suppressor), compiled in place from /root/reference; see oracle/Makefile.  TEST
// INFRASTRUCTURE ONLY; contains no algorithm.
#include "libapm/include/apm_ns.h"
extern "C" {
void* ref_apm_create(unsigned freq, int mode, int frames, int channels) {
  APM_NS* p = new APM_NS();
  if (!p->initNsModule(freq, mode, frames, channels)) { delete p; return 0; }
  return p;
}
void ref_apm_free(void* h) { delete (APM_NS*)h; }
void ref_apm_process_s16(void* h, short* data, int spc, int ch) { ((APM_NS*)h)->processCaptureStream(data, spc, ch); }
void ref_apm_process_f32(void* h, float* data, int spc, int ch) { ((APM_NS*)h)->processCaptureStream(data, spc, ch); }
}
