/*
 * ref_probe_aec.c -- glue compiled INTO oracle/_ref/libaec_ref.so next to the reference's own
 * AEC sources (compiled in place from /root/reference; see oracle/Makefile).  TEST
 * INFRASTRUCTURE ONLY; contains no algorithm.  It forces the reference's plain-C code path
 * (WebRtc_GetCPUInfoNoASM: the SSE2 overrides use a polynomial pow, aec_core_sse2.c:221-355,
 * and are not bit-equal to the C path) and runs the reference entry points the way
 * WebRtc_AMP_Port/test_aec_module.cpp:60-88 does: per 10 ms frame
 *   WebRtcAec_BufferFarend(far, 160); WebRtcAec_Process(near, 1, out, 160, delay_ms, 0).
 */
#include <stdint.h>
#include <string.h>

#include "webrtc/modules/audio_processing/aec/include/echo_cancellation.h"
#include "webrtc/system_wrappers/interface/cpu_features_wrapper.h"

void* ref_aec_create(int32_t fs) {
  void* h = NULL;
  WebRtc_GetCPUInfo = WebRtc_GetCPUInfoNoASM; /* before Create: aec_core.c:1388-1392 */
  if (WebRtcAec_Create(&h) != 0) return NULL;
  if (WebRtcAec_Init(h, fs, 48000) != 0) { /* scSampFreq as in test_aec_module.cpp:61 */
    WebRtcAec_Free(h);
    return NULL;
  }
  return h;
}

void ref_aec_free(void* h) { WebRtcAec_Free(h); }

/* frames: far/near/out [F][160] float (float-S16 range); returns the OR of the
 * return codes of WebRtcAec_Process. */
int ref_aec_run(void* h, const float* far, const float* near, float* out, int F,
                int16_t delay_ms) {
  int rc = 0;
  for (int f = 0; f < F; ++f) {
    float nbuf[160], obuf[160];
    const float* np[1] = {nbuf};
    float* op[1] = {obuf};
    memcpy(nbuf, near + (size_t)f * 160, sizeof nbuf);
    rc |= WebRtcAec_BufferFarend(h, far + (size_t)f * 160, 160);
    rc |= WebRtcAec_Process(h, np, 1, op, 160, delay_ms, 0);
    memcpy(out + (size_t)f * 160, obuf, sizeof obuf);
  }
  return rc;
}
