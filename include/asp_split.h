/*
 * asp_split.h -- C-ABI of the MI355X batched two-band QMF split / merge: the reference's
 * WebRtcSpl_AnalysisQMF / WebRtcSpl_SynthesisQMF
 * (WebRtc_AMP_Port/webrtc/common_audio/signal_processing/splitting_filter_c.c:127-212), the
 * band split AudioBuffer::SplitIntoFrequencyBands / MergeFrequencyBands use at 32 kHz
 * (modules/audio_processing/splitting_filter.cc:63-88).  Integer arithmetic, bit-exact.
 *
 * Layer 1: the reference's two functions, signature-identical (caller-owned int32[6] states);
 * each call is a batch of one channel.
 * Layer 2: AspQmfBatch_*, N independent channels per call with the four filter states of a
 * channel (TwoBandsStates, splitting_filter.h:34-45) resident in HBM.
 */
#ifndef ASP_SPLIT_H_
#define ASP_SPLIT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASP_QMF_MAX_BAND 320 /* kMaxBandFrameLength, splitting_filter_c.c:21 */

/* ---------------------------------------------------------------- layer 1 */
void WebRtcSpl_AnalysisQMF(const int16_t* in_data, int in_data_length, int16_t* low_band,
                           int16_t* high_band, int32_t* filter_state1,
                           int32_t* filter_state2); /* signal_processing_library.h:933 */
void WebRtcSpl_SynthesisQMF(const int16_t* low_band, const int16_t* high_band, int band_length,
                            int16_t* out_data, int32_t* filter_state1,
                            int32_t* filter_state2); /* signal_processing_library.h:939 */

/* ---------------------------------------------------------------- layer 2 */
typedef struct AspQmfBatch AspQmfBatch;

/* TwoBandsStates of one channel (splitting_filter.h:34-45). */
typedef struct AspQmfState {
  int32_t analysis_state1[6];
  int32_t analysis_state2[6];
  int32_t synthesis_state1[6];
  int32_t synthesis_state2[6];
} AspQmfState;

int AspQmfBatch_Create(AspQmfBatch** out, int num_channels, int device);
int AspQmfBatch_Free(AspQmfBatch* b);
int AspQmfBatch_Reset(AspQmfBatch* b); /* all states zero, as the TwoBandsStates constructor */
int AspQmfBatch_num_channels(const AspQmfBatch* b);
/* in [num_channels][2 * band_length] int16 -> low, high [num_channels][band_length];
 * band_length <= 320.  mem: 0 host, 1 device (asp_ns.h ASP_MEM_*). */
int AspQmfBatch_Analysis(AspQmfBatch* b, const int16_t* in, int band_length, int16_t* low,
                         int16_t* high, int mem);
/* low, high [num_channels][band_length] -> out [num_channels][2 * band_length]. */
int AspQmfBatch_Synthesis(AspQmfBatch* b, const int16_t* low, const int16_t* high,
                          int band_length, int16_t* out, int mem);
int AspQmfBatch_ExportState(AspQmfBatch* b, int channel, AspQmfState* out);
int AspQmfBatch_ImportState(AspQmfBatch* b, int channel, const AspQmfState* in);
int AspQmfBatch_Synchronize(AspQmfBatch* b);

/* ---- SplittingFilter: the band split / merge of AudioBuffer (splitting_filter.cc:28-170) ----
 * num_bands 2: 32 kHz <-> two 16 kHz bands (one QMF); num_bands 3: 48 kHz <-> three 16 kHz bands
 * (48 -> 64 kHz sinc resampler, QMF twice, empty top band dropped; the reverse on the way back). */
typedef struct AspSplitBatch AspSplitBatch;
int AspSplitBatch_Create(AspSplitBatch** out, int num_channels, int num_bands, int device);
int AspSplitBatch_Free(AspSplitBatch* b);
/* in [num_channels][160 * num_bands] int16 -> bands [num_bands][num_channels][160]. */
int AspSplitBatch_Analysis(AspSplitBatch* b, const int16_t* in, int16_t* bands, int mem);
int AspSplitBatch_Synthesis(AspSplitBatch* b, const int16_t* bands, int16_t* out, int mem);

#ifdef __cplusplus
}
#endif
#endif /* ASP_SPLIT_H_ */
