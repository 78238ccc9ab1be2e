/*
 * split_oracle.c -- CPU restatement of SplittingFilter::Analysis / Synthesis (WebRtc_AMP_Port/
 * webrtc/modules/audio_processing/splitting_filter.cc).  TEST INFRASTRUCTURE ONLY; parity PINNED
 * (tests/test_split_oracle.py).
 */
#include "split_oracle.h"

#include <stdlib.h>
#include <string.h>

#include "qmf_oracle.h"
#include "sinc_oracle.h"

typedef struct TwoBandsStates { /* splitting_filter.h:34-45 */
  int32_t analysis_state1[6], analysis_state2[6], synthesis_state1[6], synthesis_state2[6];
} TwoBandsStates;

struct AspSplitOracle {
  int num_bands;
  TwoBandsStates two_bands, band1, band2; /* splitting_filter.h:82-84 */
  AspSincOracle *up, *down;               /* analysis / synthesis resamplers, .cc:22-27 */
};

AspSplitOracle* asp_split_oracle_create(int num_bands) {
  AspSplitOracle* o = (AspSplitOracle*)calloc(1, sizeof *o);
  if (!o || (num_bands != 2 && num_bands != 3)) {
    free(o);
    return NULL;
  }
  o->num_bands = num_bands;
  o->up = asp_sinc_oracle_create(480, 640);
  o->down = asp_sinc_oracle_create(640, 480);
  return o;
}

void asp_split_oracle_free(AspSplitOracle* o) {
  if (!o) return;
  asp_sinc_oracle_free(o->up);
  asp_sinc_oracle_free(o->down);
  free(o);
}

void asp_split_oracle_analysis(AspSplitOracle* o, const int16_t* x, int16_t* bands) {
  if (o->num_bands == 2) { /* TwoBandsAnalysis, .cc:63-75 */
    asp_qmf_oracle_analysis(x, 320, bands, bands + 160, o->two_bands.analysis_state1,
                            o->two_bands.analysis_state2);
    return;
  }
  /* ThreeBandsAnalysis, .cc:96-131: 48 -> 64 kHz, split twice, drop the empty top band */
  int16_t buf[640], drop[160];
  asp_sinc_oracle_resample_i16(o->up, x, buf);
  asp_qmf_oracle_analysis(buf, 640, buf, buf + 320, o->two_bands.analysis_state1,
                          o->two_bands.analysis_state2);
  asp_qmf_oracle_analysis(buf, 320, bands, bands + 160, o->band1.analysis_state1,
                          o->band1.analysis_state2);
  asp_qmf_oracle_analysis(buf + 320, 320, drop, bands + 320, o->band2.analysis_state1,
                          o->band2.analysis_state2);
}

void asp_split_oracle_synthesis(AspSplitOracle* o, const int16_t* bands, int16_t* out) {
  if (o->num_bands == 2) { /* TwoBandsSynthesis, .cc:77-88 */
    asp_qmf_oracle_synthesis(bands, bands + 160, 160, out, o->two_bands.synthesis_state1,
                             o->two_bands.synthesis_state2);
    return;
  }
  /* ThreeBandsSynthesis, .cc:137-169: the uppermost band is empty (zeros) */
  int16_t buf[640];
  memset(buf, 0, sizeof buf);
  asp_qmf_oracle_synthesis(bands, bands + 160, 160, buf, o->band1.synthesis_state1,
                           o->band1.synthesis_state2);
  asp_qmf_oracle_synthesis(buf + 320, bands + 320, 160, buf + 320, o->band2.synthesis_state1,
                           o->band2.synthesis_state2);
  asp_qmf_oracle_synthesis(buf, buf + 320, 320, buf, o->two_bands.synthesis_state1,
                           o->two_bands.synthesis_state2);
  asp_sinc_oracle_resample_i16(o->down, buf, out);
}
