/*
 * sinc_oracle.c -- CPU restatement of PushSincResampler / SincResampler (WebRtc_AMP_Port/webrtc/
 * common_audio/resampler/).  TEST INFRASTRUCTURE ONLY; parity PINNED against the reference sources
 * compiled in place (tests/test_sinc_oracle.py).  "sr" = sinc_resampler.cc, "sse" =
 * sinc_resampler_sse.cc, "push" = push_sinc_resampler.cc.  Compile with -ffp-contract=off.
 */
#include "sinc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define KSIZE 32    /* kKernelSize, sinc_resampler.h:41 */
#define KOFFS 32    /* kKernelOffsetCount, sinc_resampler.h:50 */

struct AspSincOracle {
  double ratio, vsi; /* io_sample_rate_ratio_, virtual_source_idx_ */
  int request, dst_frames, block_size, r0, r3, r4, primed, first_pass;
  float kernel[KSIZE * (KOFFS + 1)];
  float* buf; /* input_buffer_: request + kKernelSize floats; r1 = 0, r2 = kKernelSize / 2 */
  const int16_t* src;
};

static void update_regions(AspSincOracle* o, int second_load) { /* sr:190-199 */
  o->r0 = second_load ? KSIZE : KSIZE / 2;
  o->r3 = o->r0 + o->request - KSIZE;
  o->r4 = o->r0 + o->request - KSIZE / 2;
  o->block_size = o->r4 - KSIZE / 2;
}

static void init_kernel(AspSincOracle* o) { /* sr:201-232, 88-101 */
  const double kAlpha = 0.16;
  const double kA0 = 0.5 * (1.0 - kAlpha), kA1 = 0.5, kA2 = 0.5 * kAlpha;
  double sinc_scale_factor = o->ratio > 1.0 ? 1.0 / o->ratio : 1.0;
  sinc_scale_factor *= 0.9;
  for (int offset_idx = 0; offset_idx <= KOFFS; ++offset_idx) {
    const float subsample_offset = (float)offset_idx / KOFFS;
    for (int i = 0; i < KSIZE; ++i) {
      const int idx = i + offset_idx * KSIZE;
      const float pre_sinc = (float)(M_PI * (i - KSIZE / 2 - subsample_offset));
      const float x = (i - subsample_offset) / KSIZE;
      const float window = (float)(kA0 - kA1 * cos(2.0 * M_PI * x) + kA2 * cos(4.0 * M_PI * x));
      o->kernel[idx] = (float)(window * ((pre_sinc == 0)
                                             ? sinc_scale_factor
                                             : (sin(sinc_scale_factor * pre_sinc) / pre_sinc)));
    }
  }
}

AspSincOracle* asp_sinc_oracle_create(int src_frames, int dst_frames) {
  AspSincOracle* o = (AspSincOracle*)calloc(1, sizeof *o);
  if (!o) return NULL;
  o->ratio = src_frames * 1.0 / dst_frames; /* push:19 */
  o->request = src_frames;
  o->dst_frames = dst_frames;
  o->buf = (float*)calloc((size_t)src_frames + KSIZE, sizeof(float));
  o->vsi = 0; /* Flush, sr:335-341 */
  o->primed = 0;
  update_regions(o, 0);
  init_kernel(o);
  o->first_pass = 1;
  return o;
}

void asp_sinc_oracle_free(AspSincOracle* o) {
  if (o) free(o->buf);
  free(o);
}

const float* asp_sinc_oracle_kernel(const AspSincOracle* o) { return o->kernel; }

static void run(AspSincOracle* o, float* destination) { /* PushSincResampler::Run, push:80-100 */
  if (o->first_pass) {
    memset(destination, 0, (size_t)o->request * sizeof(float));
    o->first_pass = 0;
    return;
  }
  for (int i = 0; i < o->request; ++i) destination[i] = (float)o->src[i];
}

/* Convolve_SSE (sse:19-57): four partial sums over i mod 4, blend, (s0 + s2) + (s1 + s3) */
static float convolve(const float* in, const float* k1, const float* k2, double factor) {
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, t[4];
  for (int i = 0; i < KSIZE; i += 4)
    for (int l = 0; l < 4; ++l) {
      s1[l] = s1[l] + in[i + l] * k1[i + l];
      s2[l] = s2[l] + in[i + l] * k2[i + l];
    }
  {
    const float f1 = (float)(1.0 - factor), f2 = (float)factor;
    for (int l = 0; l < 4; ++l) t[l] = s1[l] * f1 + s2[l] * f2;
  }
  return (t[2] + t[0]) + (t[3] + t[1]);
}

static void resample(AspSincOracle* o, int frames, float* destination) { /* sr:252-312 */
  int remaining = frames;
  if (!o->primed && remaining) {
    run(o, o->buf + o->r0);
    o->primed = 1;
  }
  while (remaining) {
    for (int i = (int)ceil((o->block_size - o->vsi) / o->ratio); i > 0; --i) {
      const int source_idx = (int)o->vsi;
      const double subsample_remainder = o->vsi - source_idx;
      const double virtual_offset_idx = subsample_remainder * KOFFS;
      const int offset_idx = (int)virtual_offset_idx;
      const float* k1 = o->kernel + offset_idx * KSIZE;
      *destination++ = convolve(o->buf + source_idx, k1, k1 + KSIZE, virtual_offset_idx - offset_idx);
      o->vsi += o->ratio;
      if (!--remaining) return;
    }
    o->vsi -= o->block_size;
    memcpy(o->buf, o->buf + o->r3, sizeof(float) * KSIZE);
    if (o->r0 == KSIZE / 2) update_regions(o, 1);
    run(o, o->buf + o->r0);
  }
}

static int16_t float_s16_to_s16(float v) { /* audio_util.h:41-49 */
  const float kMaxRound = 32767 - 0.5f, kMinRound = -32768 + 0.5f;
  if (v > 0) return v >= kMaxRound ? 32767 : (int16_t)(v + 0.5f);
  return v <= kMinRound ? -32768 : (int16_t)(v - 0.5f);
}

void asp_sinc_oracle_resample_i16(AspSincOracle* o, const int16_t* in, int16_t* out) { /* push:34-60 */
  float* tmp = (float*)malloc(sizeof(float) * (size_t)(o->dst_frames > o->request * 2 ? o->dst_frames : o->request * 2));
  o->src = in;
  if (o->first_pass) resample(o, (int)(o->block_size / o->ratio), tmp); /* ChunkSize, sr:331-333 */
  resample(o, o->dst_frames, tmp);
  for (int i = 0; i < o->dst_frames; ++i) out[i] = float_s16_to_s16(tmp[i]);
  o->src = NULL;
  free(tmp);
}
