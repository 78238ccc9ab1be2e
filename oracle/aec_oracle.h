/*
 * aec_oracle.h -- CPU restatement of the reference's WebRTC AEC path.
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; the product library never links or calls it.
 *
 * Parity: PINNED.  The reference's AEC compiles in place (oracle/Makefile ->
 * oracle/_ref/libaec_ref.so, plain-C path forced); tests/test_aec_oracle.py checks this
 * restatement against it bit for bit (live) and against tests/golden/aec_golden.npz.
 */
#ifndef ASP_AEC_ORACLE_H_
#define ASP_AEC_ORACLE_H_

#include <stdint.h>

#include "asp_aec.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct AspAecOracle AspAecOracle;

AspAecOracle* asp_aec_oracle_create(void);
void asp_aec_oracle_free(AspAecOracle* o);
int asp_aec_oracle_init(AspAecOracle* o, int32_t sampFreq, int32_t scSampFreq);
int asp_aec_oracle_set_config(AspAecOracle* o, AecConfig config);
/* WebRtcAec_enable_delay_correction on the core (aec_core.c:1876-1885): the extended filter, 32 partitions */
void asp_aec_oracle_enable_delay_correction(AspAecOracle* o, int enable);
int asp_aec_oracle_delay_correction_enabled(const AspAecOracle* o);
void asp_aec_oracle_enable_reported_delay(AspAecOracle* o, int enable);  /* core:1868-1870; 0 = delay-agnostic */
int asp_aec_oracle_reported_delay_enabled(const AspAecOracle* o);
int asp_aec_oracle_get_delay_metrics(AspAecOracle* o, int* median, int* std); /* ec:550-571, core:1780-1836 */
void asp_aec_oracle_export_delay(const AspAecOracle* o, AspAecDelayState* d);
/* the delay estimator on its own (utility/delay_estimator.c, delay_estimator_wrapper.c; robust validation on,
 * 125 blocks of history): the seam utility/delay_estimator_unittest.cc tests */
void asp_de_oracle_init(AspAecDelayState* d, int lookahead, int allowed_offset);
void asp_de_oracle_add_binary_far(AspAecDelayState* d, uint32_t binary_far);  /* WebRtc_AddBinaryFarSpectrum */
int asp_de_oracle_process_binary(AspAecDelayState* d, uint32_t binary_near);  /* WebRtc_ProcessBinarySpectrum */
void asp_de_oracle_add_far(AspAecDelayState* d, const float* far_spectrum);   /* WebRtc_AddFarSpectrumFloat */
int asp_de_oracle_process(AspAecDelayState* d, const float* near_spectrum);   /* WebRtc_DelayEstimatorProcessFloat */
float asp_de_oracle_quality(const AspAecDelayState* d);                       /* WebRtc_last_delay_quality */
void asp_aec_oracle_export_skew(const AspAecOracle* o, float* position, float* skew, int* resample, int* index);
int asp_aec_oracle_buffer_farend(AspAecOracle* o, const float* farend, int nrOfSamples);
int asp_aec_oracle_process(AspAecOracle* o, const float* nearend, float* out, int nrOfSamples,
                           int msInSndCardBuf, int32_t skew);
/* WebRtcAec_Process with two bands (32 kHz): nearendH / outH are the 8-16 kHz band. */
int asp_aec_oracle_process_bands(AspAecOracle* o, const float* nearend, const float* nearendH,
                                 float* out, float* outH, int nrOfSamples, int msInSndCardBuf,
                                 int32_t skew);
int asp_aec_oracle_echo_status(const AspAecOracle* o);
int asp_aec_oracle_error_code(const AspAecOracle* o);
void asp_aec_oracle_export_metrics(const AspAecOracle* o, AspAecMetricsState* m);
int asp_aec_oracle_get_metrics(const AspAecOracle* o, AecMetrics* metrics); /* ec:456-548 */
void asp_aec_oracle_export(const AspAecOracle* o, AspAecState* st, AspAecControl* ctl);
void asp_aec_oracle_import(AspAecOracle* o, const AspAecState* st);
/* test_aec_module.cpp:75-88 for one stream: F x (BufferFarend(far, n) + Process(near, n, delay)).
 * far / near / out [F][n].  Returns the OR of the return codes. */
int asp_aec_oracle_run(AspAecOracle* o, const float* far, const float* near, float* out, int F,
                       int n, int delay_ms);
/* S independent streams on `threads` pthreads; frames [F][S][n].  bench.py's cpu_baseline leg. */
int asp_aec_oracle_run_mt(int num_streams, const float* far, const float* near, float* out, int F,
                          int n, int delay_ms, int32_t fs, int threads);

void asp_aec_oracle_rdft128(float* a, int isgn); /* aec_rdft.c:539-556 */
/* which: 0 rdft_w[64], 1 rdft_wk3ri_first[16], 2 rdft_wk3ri_second[16], 3 sqrtHanning[65],
 * 4 weightCurve[65], 5 overDriveCurve[65]. */
const float* asp_aec_oracle_table(int which);

#ifdef __cplusplus
}
#endif
#endif /* ASP_AEC_ORACLE_H_ */
