/*
 * bt_main.c -- WAV -> block-thresholding denoiser -> WAV driver in plain C.
 *
 * Restates the loop of the reference's Denoise/BlockThresholding/main.cpp:44-116 over this library's
 * drop-in blockThreshold_* entry points (include/asp_bt.h).  The reference takes {infile, outfile,
 * frame_ms} from a JSON file at a hard-coded Windows path through jansson (main.cpp:12-42; jansson
 * ships as a Windows .lib only), here they are command-line arguments; everything else is the
 * reference's behaviour, quirks included:
 *   - header copied verbatim (main.cpp:53-54);
 *   - num_samples = data.size / channels / bits_per_sample / 8 (main.cpp:61: divides by the bits AND by
 *     8) is only compared with the frame size and never counted down: a file whose data chunk is
 *     shorter than channels * bits * 8 * frame_size bytes is copied as a bare header, any other file
 *     is read to its end;
 *   - frame = half a window of int16 samples read straight from the data chunk, so a stereo file is
 *     consumed as interleaved mono (wav_io never de-interleaves);
 *   - blockThreshold_denoise_int16 per frame; on MARS_CAN_OUTPUT (every 8th call) one macroblock is
 *     fetched with blockThreshold_output_int16 and written (main.cpp:94-108);
 *   - a short final read ends the run: blockThreshold_flush_int16, then the partial frame is written
 *     back unprocessed (main.cpp:83-91).
 *
 *   bt_main in.wav out.wav frame_ms [-q]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asp_bt.h"
#include "wav_io.h"

int main(int argc, char* argv[]) {
  if (argc < 4) {
    fprintf(stderr, "usage: bt_main in.wav out.wav frame_ms [-q]\n");
    return 1;
  }
  const int quiet = argc > 4 && strcmp(argv[4], "-q") == 0;
  const int32_t time_win = (int32_t)atoi(argv[3]);
  FILE* in_file = fopen(argv[1], "rb");
  FILE* out_file = fopen(argv[2], "wb");
  if (!in_file || !out_file) {
    fprintf(stderr, "error: can not open file!!!\n");
    return 1;
  }
  WAV_HEADER header;
  if (read_header(&header, in_file) != 0) {
    fprintf(stderr, "error: can not read the wav header\n");
    return 1;
  }
  write_header(&header, out_file);

  int32_t ret = MARS_OK;
  const int32_t freq = (int32_t)header.format.sample_per_sec;
  const int32_t channels = header.format.channels;
  const int32_t bits_per_sample = header.format.bits_per_sample;
  if (channels <= 0 || bits_per_sample <= 0) {
    fprintf(stderr, "error: bad wav format\n");
    return 1;
  }
  const int32_t num_samples = (int32_t)(header.data.size / channels / bits_per_sample / 8); /* main.cpp:61 */
  MarsBlockThreshold_t* handle = blockThreshold_init(time_win, freq, &ret);
  if (ret != MARS_OK || !handle) {
    fprintf(stderr, "error: blockThreshold_init (%d ms at %d Hz): this build has windows of up to 2048 "
                    "samples (128 ms at 16 kHz, 42 ms at 48 kHz)\n", (int)time_win, (int)freq);
    return -1;
  }
  const int32_t frame_size = blockThreshold_samples_per_time(handle);
  const int32_t outbuf_len = blockThreshold_max_output(handle);
  int16_t* inbuf = (int16_t*)malloc(sizeof(int16_t) * (size_t)frame_size);
  int16_t* outbuf = (int16_t*)malloc(sizeof(int16_t) * (size_t)outbuf_len);
  if (!inbuf || !outbuf) {
    fprintf(stderr, "Error in malloc!!!\n");
    return -1;
  }

  int32_t frm_cnt = 0;
  while (num_samples > frame_size) {
    const int32_t readed = (int32_t)fread(inbuf, sizeof(int16_t), (size_t)frame_size, in_file);
    if (readed != frame_size) {
      if (!quiet) fprintf(stdout, "end of file, flush sample in internal buffer!!!\n");
      const int32_t out_size = blockThreshold_flush_int16(handle, outbuf, outbuf_len);
      if (out_size > 0) fwrite(outbuf, sizeof(int16_t), (size_t)out_size, out_file);
      fwrite(inbuf, sizeof(int16_t), (size_t)readed, out_file);
      break;
    }
    if (!quiet) fprintf(stdout, "Frame: %d\n", frm_cnt);
    frm_cnt++;
    ret = blockThreshold_denoise_int16(handle, inbuf, frame_size);
    if (ret == MARS_ERROR_PARAMS) {
      fprintf(stderr, "ret = MARS_ERROR_PARAMS\n");
      break;
    } else if (ret == MARS_NEED_MORE_SAMPLES) {
      continue;
    } else if (ret == MARS_CAN_OUTPUT) {
      const int32_t len = blockThreshold_output_int16(handle, outbuf, outbuf_len);
      fwrite(outbuf, sizeof(int16_t), (size_t)len, out_file);
      if (!quiet) fprintf(stdout, "one macro block processed!!\n");
    }
  }

  free(inbuf);
  free(outbuf);
  fclose(in_file);
  fclose(out_file);
  blockThreshold_free(handle);
  return 0;
}
