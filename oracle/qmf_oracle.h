/*
 * qmf_oracle.h -- CPU restatement of the reference's two-band QMF split / merge
 * (common_audio/signal_processing/splitting_filter_c.c).  TEST INFRASTRUCTURE ONLY.
 * Parity: PINNED -- the reference file compiles in place (oracle/_ref/libspl_ref.so) and
 * tests/test_qmf_oracle.py checks this restatement against it bit for bit.
 */
#ifndef ASP_QMF_ORACLE_H_
#define ASP_QMF_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void asp_qmf_oracle_analysis(const int16_t* in_data, int in_data_length, int16_t* low_band,
                             int16_t* high_band, int32_t* filter_state1, int32_t* filter_state2);
void asp_qmf_oracle_synthesis(const int16_t* low_band, const int16_t* high_band, int band_length,
                              int16_t* out_data, int32_t* filter_state1, int32_t* filter_state2);
#ifdef __cplusplus
}
#endif
#endif
