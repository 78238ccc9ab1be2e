/*
 * test_aec_module.c -- mic.wav + speaker.wav -> echo canceller -> WAV driver in plain C.
 *
 * Restates the loop of the reference's WebRtc_AMP_Port/test_aec_module.cpp:13-106 for 8 / 16 kHz
 * mono over this library's drop-in WebRtcAec_* entry points (include/asp_aec.h): the mic header
 * is copied verbatim (:47), WebRtcAec_Init(handle, fs, 48000) (:61), per 10 ms frame the far end
 * is buffered first and the capture frame processed in place with stream_delay_ms = 0 and no
 * drift (:67-88), int16 -> float-S16 is value preserving (channel_buffer.cc:43-53) and the result
 * goes back through FloatS16ToS16 rounding (audio_util.h:41-49).  The `while (!feof)` loop also
 * processes the final short reads with their stale tails, like the reference.
 *
 *   test_aec_module mic.wav speaker.wav aec_result.wav [-q]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asp_aec.h"
#include "wav_io.h"

static int16_t float_s16_to_s16(float v) { /* audio_util.h:41-49 */
  const float kMaxRound = 32767 - 0.5f, kMinRound = -32768 + 0.5f;
  if (v > 0) return v >= kMaxRound ? 32767 : (int16_t)(v + 0.5f);
  return v <= kMinRound ? -32768 : (int16_t)(v - 0.5f);
}

int main(int argc, char* argv[]) {
  if (argc < 4) {
    printf("Usage: %s mic.wav speaker.wav aec_result.wav\n", argv[0]);
    return -1;
  }
  const int quiet = argc > 4 && strcmp(argv[4], "-q") == 0;
  FILE* mic_file = fopen(argv[1], "rb");
  FILE* speaker_file = fopen(argv[2], "rb");
  FILE* result_file = fopen(argv[3], "wb");
  if (!mic_file || !speaker_file || !result_file) {
    printf("Fail to open file !!!\n");
    return -1;
  }
  WAV_HEADER mic_header, speaker_header;
  if (read_header(&mic_header, mic_file) != 0) {
    printf("Fail to parse wav file: %s\n", argv[1]);
    return -1;
  }
  if (read_header(&speaker_header, speaker_file) != 0) {
    printf("Fail to parse wav file: %s\n", argv[2]);
    return -1;
  }
  if (mic_header.format.bits_per_sample != 16 || speaker_header.format.bits_per_sample != 16) {
    printf("Now only support 16 bits per sample!\n");
    return -1;
  }
  if (mic_header.format.sample_per_sec != speaker_header.format.sample_per_sec ||
      mic_header.format.channels != 1 || speaker_header.format.channels != 1) {
    printf("mic and speaker must be mono at the same rate\n");
    return -1;
  }
  write_header(&mic_header, result_file);

  const uint32_t frequency = (uint32_t)mic_header.format.sample_per_sec;
  const int length = (int)(frequency / 100);
  int16_t* mic_buf = (int16_t*)calloc((size_t)length, sizeof(int16_t));
  int16_t* speaker_buf = (int16_t*)calloc((size_t)length, sizeof(int16_t));
  int16_t* result_buf = (int16_t*)calloc((size_t)length, sizeof(int16_t));
  float* far_band = (float*)calloc((size_t)length, sizeof(float));
  float* near_band = (float*)calloc((size_t)length, sizeof(float));

  void* aec = NULL;
  if (WebRtcAec_Create(&aec) != 0) {
    printf("WebRtcAec_Create failed\n");
    return -1;
  }
  if (WebRtcAec_Init(aec, (int32_t)frequency, 48000) != 0) {
    printf("WebRtcAec_Init(%u) failed: error %d\n", frequency, WebRtcAec_get_error_code(aec));
    return -1;
  }
  int32_t frm_cnt = 0;
  const int16_t stream_delay_ms = 0;
  const int32_t stream_drift_samples = 0;
  while (!feof(mic_file) && !feof(speaker_file)) {
    read_samples(speaker_buf, length, &speaker_header, speaker_file);
    for (int i = 0; i < length; ++i) far_band[i] = (float)speaker_buf[i];
    WebRtcAec_BufferFarend(aec, far_band, (int16_t)length);

    read_samples(mic_buf, length, &mic_header, mic_file);
    for (int i = 0; i < length; ++i) near_band[i] = (float)mic_buf[i];
    const float* in_bands[1] = {near_band};
    float* out_bands[1] = {near_band}; /* in place, like the reference driver */
    WebRtcAec_Process(aec, in_bands, 1, out_bands, (int16_t)length, stream_delay_ms,
                      stream_drift_samples);
    for (int i = 0; i < length; ++i) result_buf[i] = float_s16_to_s16(near_band[i]);
    write_samples(result_buf, length, &mic_header, result_file);
    if (!quiet) printf("Frame #%d\n", frm_cnt);
    frm_cnt++;
  }
  printf("%d frames\n", frm_cnt);
  WebRtcAec_Free(aec);
  fclose(mic_file);
  fclose(speaker_file);
  fclose(result_file);
  free(mic_buf);
  free(speaker_buf);
  free(result_buf);
  free(far_band);
  free(near_band);
  return 0;
}
