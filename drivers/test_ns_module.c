/*
 * test_ns_module.c -- WAV -> noise suppressor -> WAV driver in plain C.
 *
 * Restates the loop of the reference's WebRtc_AMP_Port/test_ns_module.cpp:23-122
 * for 16 kHz mono over this library's drop-in WebRtcNs_* entry points
 * (include/asp_ns.h), i.e. the GPU does the spectral work, the host stays C:
 *   header copied verbatim (test_ns_module.cpp:44-51), policy kModerate = 1
 *   (:70), `while (!feof)` frame loop that also processes the final short read
 *   with a stale tail (:83-86), int16 -> float-S16 value-preserving
 *   (channel_buffer.cc:43-53), Analyze then in-place Process (:97-99),
 *   FloatS16ToS16 rounding (audio_util.h:41-49), write 160 samples (:106).
 *
 *   test_ns_module in.wav out.wav
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asp_ns.h"
#include "wav_io.h"

static int16_t float_s16_to_s16(float v) { /* audio_util.h:41-49 */
  const float kMaxRound = 32767 - 0.5f, kMinRound = -32768 + 0.5f;
  if (v > 0) return v >= kMaxRound ? 32767 : (int16_t)(v + 0.5f);
  return v <= kMinRound ? -32768 : (int16_t)(v - 0.5f);
}

int main(int argc, char* argv[]) {
  if (argc < 3) {
    printf("Please input parameters\n");
    return -1;
  }
  const int quiet = argc > 3 && strcmp(argv[3], "-q") == 0;
  printf("Process %s -> %s\n", argv[1], argv[2]);
  FILE* fr = fopen(argv[1], "rb");
  FILE* fw = fopen(argv[2], "wb");
  if (!fr || !fw) {
    printf("Can't open file!\n");
    return -1;
  }
  WAV_HEADER header;
  if (read_header(&header, fr) != 0) {
    printf("Fail to read wav header!\n");
    return -1;
  }
  if (!quiet) print_header(&header);
  write_header(&header, fw);
  if (header.format.bits_per_sample != 16) {
    printf("Now only support 16 bits per sample!\n");
    return -1;
  }
  const uint32_t frequency = (uint32_t)header.format.sample_per_sec;
  const int length = (int)(frequency / 100);
  /* This driver feeds ONE band of ONE channel per 10 ms (the 8 and 16 kHz mono cases of
   * test_ns_module.cpp:59-113: 80 / 160 samples per frame).  At 32 / 48 kHz the reference driver goes through
   * AudioBuffer::SplitIntoFrequencyBands / MergeFrequencyBands (test_ns_module.cpp:92-104): that
   * chain is include/apm_ns.h (APM_NS at 32 / 48 kHz) and drivers/apm_ns_raw.cpp here, so other
   * rates and channel counts are refused instead of being processed wrongly. */
  if ((frequency != 16000 && frequency != 8000) || header.format.channels != 1) {
    printf("test_ns_module: %u Hz / %d channel(s) not supported by this driver (8000 / 16000 Hz mono only; "
           "use drivers/apm_ns_raw for 32 / 48 kHz or multi-channel input)\n",
           frequency, (int)header.format.channels);
    return 2;
  }
  NsHandle* handle = NULL;
  if (WebRtcNs_Create(&handle) != 0) {
    printf("WebRtcNs_Create failed: %s\n", AspNs_last_error());
    return -1;
  }
  if (WebRtcNs_Init(handle, frequency) != 0) {
    printf("WebRtcNs_Init(%u) failed: %s\n", frequency, AspNs_last_error());
    return -1;
  }
  WebRtcNs_set_policy(handle, 1 /* kModerate */);

  int16_t* input = (int16_t*)calloc((size_t)length, sizeof(int16_t));
  int16_t* output = (int16_t*)calloc((size_t)length, sizeof(int16_t));
  float* band = (float*)calloc((size_t)length, sizeof(float));
  int32_t frm_cnt = 0;
  while (!feof(fr)) {
    read_samples(input, length, &header, fr);
    for (int i = 0; i < length; ++i) band[i] = (float)input[i];
    const float* in_bands[1] = {band};
    float* out_bands[1] = {band}; /* in place, like the reference driver */
    WebRtcNs_Analyze(handle, band);
    WebRtcNs_Process(handle, in_bands, 1, out_bands);
    /* `input` keeps the samples just read: a short final read re-processes the stale tail */
    for (int i = 0; i < length; ++i) output[i] = float_s16_to_s16(band[i]);
    write_samples(output, length, &header, fw);
    if (!quiet) printf("Frame #%d\n", frm_cnt);
    frm_cnt++;
  }
  printf("%d frames\n", frm_cnt);
  WebRtcNs_Free(handle);
  fclose(fr);
  fclose(fw);
  free(input);
  free(output);
  free(band);
  return 0;
}
