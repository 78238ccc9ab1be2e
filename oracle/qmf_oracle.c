/*
 * qmf_oracle.c -- CPU restatement of WebRtcSpl_AnalysisQMF / WebRtcSpl_SynthesisQMF
 * (WebRtc_AMP_Port/webrtc/common_audio/signal_processing/splitting_filter_c.c).  TEST
 * INFRASTRUCTURE ONLY; parity PINNED against the reference file compiled in place
 * (tests/test_qmf_oracle.py).  Integer arithmetic throughout.
 */
#include "qmf_oracle.h"

static const uint16_t kAllPass1[3] = {6418, 36982, 57261};  /* splitting_filter_c.c:25 */
static const uint16_t kAllPass2[3] = {21333, 49062, 63010}; /* splitting_filter_c.c:26 */

static int32_t sub_sat(int32_t a, int32_t b) { /* WebRtcSpl_SubSatW32, spl_inl.h:57-74 */
  int32_t d = (int32_t)((uint32_t)a - (uint32_t)b);
  if (a < 0) {
    if (b > 0 && d > 0) d = (int32_t)0x80000000;
  } else {
    if (b < 0 && d < 0) d = 0x7FFFFFFF;
  }
  return d;
}

/* WEBRTC_SPL_SCALEDIFF32(A, B, C) = C + (B >> 16) * A + (((uint32_t)(0xFFFF & B) * A) >> 16)
 * (signal_processing_library.h:77-79): the sum is formed in unsigned arithmetic. */
static int32_t scale_diff(uint16_t a, int32_t b, int32_t c) {
  const uint32_t hi = (uint32_t)((b >> 16) * (int32_t)a);
  const uint32_t lo = ((uint32_t)(0x0000FFFF & b) * (uint32_t)a) >> 16;
  return (int32_t)((uint32_t)c + hi + lo);
}

static int16_t sat16(int32_t v) { /* WebRtcSpl_SatW32ToW16, spl_inl.h:27-36 */
  return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v);
}

/* WebRtcSpl_AllPassQMF (splitting_filter_c.c:45-125): three first-order all-pass sections in
 * cascade; the three passes over the vector are fused into one loop over time (each section at
 * time k needs only its input at k and k-1 and its own output at k-1). */
static void all_pass(const int32_t* in, int n, int32_t* out, const uint16_t* a, int32_t* st) {
  int32_t x1 = st[0], y1a = st[1], y1b = st[2], y2a = st[3], y2b = st[4], y3 = st[5];
  for (int k = 0; k < n; ++k) {
    const int32_t x = in[k];
    const int32_t y1 = scale_diff(a[0], sub_sat(x, y1a), x1);
    const int32_t y2 = scale_diff(a[1], sub_sat(y1, y2a), y1b);
    const int32_t y = scale_diff(a[2], sub_sat(y2, y3), y2b);
    x1 = x;
    y1a = y1;
    y1b = y1;
    y2a = y2;
    y2b = y2;
    y3 = y;
    out[k] = y;
  }
  st[0] = x1;
  st[1] = y1a;
  st[2] = y1b;
  st[3] = y2a;
  st[4] = y2b;
  st[5] = y3;
}

void asp_qmf_oracle_analysis(const int16_t* in_data, int in_data_length, int16_t* low_band,
                             int16_t* high_band, int32_t* filter_state1, int32_t* filter_state2) {
  int32_t half_in1[320] = {0}, half_in2[320] = {0}, filter1[320], filter2[320];
  const int band_length = in_data_length / 2;
  for (int i = 0, k = 0; i < band_length; i++, k += 2) { /* splitting_filter_c.c:145-149 */
    half_in2[i] = (int32_t)((uint32_t)(int32_t)in_data[k] << 10);
    half_in1[i] = (int32_t)((uint32_t)(int32_t)in_data[k + 1] << 10);
  }
  all_pass(half_in1, band_length, filter1, kAllPass1, filter_state1);
  all_pass(half_in2, band_length, filter2, kAllPass2, filter_state2);
  for (int i = 0; i < band_length; i++) { /* :159-166 */
    int32_t tmp = (filter1[i] + filter2[i] + 1024) >> 11;
    low_band[i] = sat16(tmp);
    tmp = (filter1[i] - filter2[i] + 1024) >> 11;
    high_band[i] = sat16(tmp);
  }
}

void asp_qmf_oracle_synthesis(const int16_t* low_band, const int16_t* high_band, int band_length,
                              int16_t* out_data, int32_t* filter_state1, int32_t* filter_state2) {
  int32_t half_in1[320] = {0}, half_in2[320] = {0}, filter1[320], filter2[320];
  for (int i = 0; i < band_length; i++) { /* :184-190 */
    int32_t tmp = (int32_t)low_band[i] + (int32_t)high_band[i];
    half_in1[i] = (int32_t)((uint32_t)tmp << 10);
    tmp = (int32_t)low_band[i] - (int32_t)high_band[i];
    half_in2[i] = (int32_t)((uint32_t)tmp << 10);
  }
  all_pass(half_in1, band_length, filter1, kAllPass2, filter_state1);
  all_pass(half_in2, band_length, filter2, kAllPass1, filter_state2);
  for (int i = 0, k = 0; i < band_length; i++) { /* :201-210 */
    int32_t tmp = (filter2[i] + 512) >> 10;
    out_data[k++] = sat16(tmp);
    tmp = (filter1[i] + 512) >> 10;
    out_data[k++] = sat16(tmp);
  }
}
