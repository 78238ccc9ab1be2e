/*
 * split_oracle.h -- CPU restatement of the reference's SplittingFilter (modules/audio_processing/
 * splitting_filter.cc:28-170): the two-band (32 kHz) and three-band (48 kHz) split / merge of
 * AudioBuffer, composed from qmf_oracle.c and sinc_oracle.c.  TEST INFRASTRUCTURE ONLY.
 * Parity: PINNED against the reference compiled in place (oracle/_ref/libsplit_ref.so,
 * tests/test_split_oracle.py).
 */
#ifndef ASP_SPLIT_ORACLE_H_
#define ASP_SPLIT_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct AspSplitOracle AspSplitOracle;
AspSplitOracle* asp_split_oracle_create(int num_bands); /* 2 or 3 */
void asp_split_oracle_free(AspSplitOracle* o);
/* x [160 * num_bands] -> bands [num_bands][160] */
void asp_split_oracle_analysis(AspSplitOracle* o, const int16_t* x, int16_t* bands);
void asp_split_oracle_synthesis(AspSplitOracle* o, const int16_t* bands, int16_t* out);
#ifdef __cplusplus
}
#endif
#endif
