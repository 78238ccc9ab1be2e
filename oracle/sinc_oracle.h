/*
 * sinc_oracle.h -- CPU restatement of the reference's push sinc resampler
 * (common_audio/resampler/{push_sinc_resampler,sinc_resampler,sinc_resampler_sse}.cc) as the
 * three-band split uses it (48 <-> 64 kHz, int16 in / out).  TEST INFRASTRUCTURE ONLY.
 * Parity: PINNED -- the reference sources compile in place (oracle/_ref/libsinc_ref.so) and
 * tests/test_sinc_oracle.py checks this restatement against them bit for bit.
 */
#ifndef ASP_SINC_ORACLE_H_
#define ASP_SINC_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct AspSincOracle AspSincOracle;
AspSincOracle* asp_sinc_oracle_create(int src_frames, int dst_frames);
void asp_sinc_oracle_free(AspSincOracle* o);
/* PushSincResampler::Resample(const int16_t*, ...): src_frames in -> dst_frames out. */
void asp_sinc_oracle_resample_i16(AspSincOracle* o, const int16_t* in, int16_t* out);
/* The 33 x 32 kernel table (SincResampler::InitializeKernel). */
const float* asp_sinc_oracle_kernel(const AspSincOracle* o);
#ifdef __cplusplus
}
#endif
#endif
