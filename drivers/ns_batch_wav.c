/*
 * ns_batch_wav.c -- many WAV files through ONE batched noise suppressor (plain C host code over
 * the batched C-ABI, int16 PCM straight into the fused HIP step).
 *
 *   ns_batch_wav out_dir in1.wav in2.wav ...      (16 kHz, 16-bit; each file = one stream)
 *
 * Per file it reproduces what the reference's single-stream driver
 * (WebRtc_AMP_Port/test_ns_module.cpp:44-109) would have written: header copied verbatim,
 * policy 1, `while (!feof)` framing -- a short final read keeps the stale tail of the previous
 * frame and an exact multiple of 160 samples yields one extra stale frame (:83-86) -- and
 * FloatS16ToS16 rounding.  Streams shorter than the longest are fed zeros afterwards; their
 * outputs are cut at their own frame count.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asp_ns.h"
#include "wav_io.h"

#define FRAME 160
#define CHUNK 100 /* frames per batched call */

typedef struct {
  WAV_HEADER header;
  int16_t* pcm;
  long samples, frames;
  const char* path;
} Stream;

static const char* base_name(const char* p) {
  const char* s = strrchr(p, '/');
  return s ? s + 1 : p;
}

int main(int argc, char* argv[]) {
  if (argc < 3) {
    printf("usage: ns_batch_wav out_dir in1.wav [in2.wav ...]\n");
    return -1;
  }
  const int S = argc - 2;
  Stream* st = (Stream*)calloc((size_t)S, sizeof(Stream));
  long max_frames = 0;
  for (int s = 0; s < S; ++s) {
    FILE* fr = fopen(argv[2 + s], "rb");
    if (!fr || read_header(&st[s].header, fr) != 0) {
      printf("Fail to read wav header: %s\n", argv[2 + s]);
      return -1;
    }
    if (st[s].header.format.bits_per_sample != 16 || st[s].header.format.sample_per_sec != 16000) {
      printf("%s: only 16 kHz / 16-bit input is supported\n", argv[2 + s]);
      return -1;
    }
    long cap = 1 << 16, n = 0;
    int16_t* buf = (int16_t*)malloc(sizeof(int16_t) * (size_t)cap);
    for (;;) {
      if (n + FRAME > cap) buf = (int16_t*)realloc(buf, sizeof(int16_t) * (size_t)(cap *= 2));
      int got = read_samples(buf + n, FRAME, &st[s].header, fr);
      n += got;
      if (got < FRAME) break;
    }
    fclose(fr);
    st[s].pcm = buf;
    st[s].samples = n;
    st[s].frames = n / FRAME + 1; /* the reference loop runs once more after the last full read */
    st[s].path = argv[2 + s];
    if (st[s].frames > max_frames) max_frames = st[s].frames;
  }

  AspNsBatch* b = NULL;
  if (AspNsBatch_Create(&b, S, 0) != ASP_OK || AspNsBatch_Init(b, 16000) != ASP_OK ||
      AspNsBatch_set_policy(b, 1) != ASP_OK) {
    printf("cannot create the GPU batch: %s\n", AspNs_last_error());
    return -1;
  }
  int16_t* in = (int16_t*)calloc((size_t)CHUNK * S * FRAME, sizeof(int16_t));
  int16_t* out = (int16_t*)calloc((size_t)CHUNK * S * FRAME, sizeof(int16_t));
  int16_t* last = (int16_t*)calloc((size_t)S * FRAME, sizeof(int16_t)); /* per-stream read buffer */
  int16_t** res = (int16_t**)calloc((size_t)S, sizeof(int16_t*));
  for (int s = 0; s < S; ++s) res[s] = (int16_t*)malloc(sizeof(int16_t) * (size_t)st[s].frames * FRAME);

  for (long f0 = 0; f0 < max_frames; f0 += CHUNK) {
    const int nf = (int)(max_frames - f0 < CHUNK ? max_frames - f0 : CHUNK);
    for (int f = 0; f < nf; ++f)
      for (int s = 0; s < S; ++s) {
        int16_t* dst = in + ((size_t)f * S + s) * FRAME;
        const long pos = (f0 + f) * FRAME;
        if (f0 + f < st[s].frames) {
          long avail = st[s].samples - pos;
          if (avail > FRAME) avail = FRAME;
          if (avail > 0) memcpy(last + (size_t)s * FRAME, st[s].pcm + pos, sizeof(int16_t) * (size_t)avail);
          memcpy(dst, last + (size_t)s * FRAME, sizeof(int16_t) * FRAME); /* stale tail kept */
        } else {
          memset(dst, 0, sizeof(int16_t) * FRAME);
        }
      }
    if (AspNsBatch_AnalyzeProcessS16(b, in, out, nf, ASP_MEM_HOST) != ASP_OK) {
      printf("AspNsBatch_AnalyzeProcessS16: %s\n", AspNs_last_error());
      return -1;
    }
    for (int f = 0; f < nf; ++f)
      for (int s = 0; s < S; ++s)
        if (f0 + f < st[s].frames)
          memcpy(res[s] + (f0 + f) * FRAME, out + ((size_t)f * S + s) * FRAME, sizeof(int16_t) * FRAME);
  }
  for (int s = 0; s < S; ++s) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", argv[1], base_name(st[s].path));
    FILE* fw = fopen(path, "wb");
    if (!fw) {
      printf("Can't open %s\n", path);
      return -1;
    }
    write_header(&st[s].header, fw);
    write_samples(res[s], (int)(st[s].frames * FRAME), &st[s].header, fw);
    fclose(fw);
  }
  printf("%d streams, %ld frames (longest)\n", S, max_frames);
  AspNsBatch_Free(b);
  return 0;
}
