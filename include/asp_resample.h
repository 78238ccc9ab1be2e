/*
 * asp_resample.h -- C-ABI of the MI355X batched push sinc resampler: the reference's
 * PushSincResampler / SincResampler (WebRtc_AMP_Port/webrtc/common_audio/resampler/
 * push_sinc_resampler.cc:16-100, sinc_resampler.cc:150-355, sinc_resampler_sse.cc:19-57) as the
 * three-band split uses it for 48 <-> 64 kHz (modules/audio_processing/splitting_filter.cc:91-170),
 * int16 in / int16 out, N independent channels per call.
 *
 * The resampler's position arithmetic (virtual_source_idx_ in double, region bookkeeping,
 * priming) depends only on the call sequence, so it runs once per call on the host and becomes a
 * table of (source index, kernel offset, blend factors) per output sample; the 32-tap convolutions
 * run on the GPU with the four-partial-sum order of the SSE kernel an x86-64 build of the
 * reference uses, so outputs are bit-identical to it.
 */
#ifndef ASP_RESAMPLE_H_
#define ASP_RESAMPLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct AspSincBatch AspSincBatch;

/* PushSincResampler(source_frames, destination_frames) for every channel. */
int AspSincBatch_Create(AspSincBatch** out, int num_channels, int source_frames,
                        int destination_frames, int device);
int AspSincBatch_Free(AspSincBatch* b);
int AspSincBatch_num_channels(const AspSincBatch* b);
/* PushSincResampler::Resample(const int16_t*, ...): in [num_channels][source_frames] ->
 * out [num_channels][destination_frames].  mem: 0 host, 1 device (asp_ns.h ASP_MEM_*). */
int AspSincBatch_Resample(AspSincBatch* b, const int16_t* in, int16_t* out, int mem);
int AspSincBatch_Synchronize(AspSincBatch* b);
/* Run the batch on the caller's HIP stream (hipStream_t as void*) instead of its own. */
int AspSincBatch_SetStream(AspSincBatch* b, void* hip_stream);
/* The 33 x 32 kernel table built on the host (SincResampler::InitializeKernel), for the parity
 * tests.  Returns the number of floats written (1056). */
int AspSincBatch_kernel_table(const AspSincBatch* b, float* out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* ASP_RESAMPLE_H_ */
