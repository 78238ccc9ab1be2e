// aec_estimator.h -- the AEC's binary delay estimator and the delay-agnostic mode's per-stream far-buffer control as
// device functions of one wave64 per stream (utility/delay_estimator.c, delay_estimator_wrapper.c, aec_core.c:797-850,
// 1696-1751), shared by the estimator kernels (aec_delay_kernels.hip) and the process kernel's fused delay-agnostic
// form (aec_kernels.hip).  Lane q owns entries q and q + 64 of the 125 / 126-entry arrays, in registers; the
// estimator's scalars live in SGPRs.  No LDS.  Integer and float operations are the reference's, in its order.
#pragma once
#include <hip/hip_runtime.h>

#include "aec_binspec.h"
#include "aec_layout.h"

namespace aspaec_est {
using namespace aspaec;

constexpr int kHist = ASP_AEC_DELAY_HISTORY;        // 125
constexpr int kNearHist = ASP_AEC_DELAY_HISTORY + 1;  // max_lookahead = kHistorySizeBlocks (aec_core.c:1363-1366)
constexpr int kMaxBitCountsQ9 = 32 << 9;              // delay_estimator.h:17

// ring_buffer.c position logic on the stream's far buffer (250 partitions)
struct FarPos {
  int read, write, wrap;
};
__device__ __forceinline__ int fp_avail_read(const FarPos& r) {  // ring_buffer.c:231-240
  return r.wrap == 0 ? r.write - r.read : kFarSlots - r.read + r.write;
}
__device__ __forceinline__ int fp_move_read(FarPos& r, int n) {  // WebRtc_MoveReadPtr, ring_buffer.c:195-228
  const int readable = fp_avail_read(r), free_elements = kFarSlots - readable;
  int pos = r.read;
  if (n > readable) n = readable;
  if (n < -free_elements) n = -free_elements;
  pos += n;
  if (pos > kFarSlots) {
    pos -= kFarSlots;
    r.wrap = 0;
  }
  if (pos < 0) {
    pos += kFarSlots;
    r.wrap = 1;
  }
  r.read = pos;
  return n;
}
__device__ __forceinline__ void fp_write_one(FarPos& r) {  // WebRtc_WriteBuffer of one element, ring_buffer.c:161-192
  const int free_elements = kFarSlots - fp_avail_read(r);
  const int write_elements = free_elements < 1 ? free_elements : 1;
  int m = write_elements;
  const int margin = kFarSlots - r.write;
  if (write_elements > margin) {
    r.write = 0;
    m -= margin;
    r.wrap = 1;
  }
  r.write += m;
}

// ---- the estimator's arrays in registers: lane q holds entries q and q + 64 of each 125 / 126-entry array
struct Hist {
  unsigned fh0, fh1;  // binary_far_history   (125)
  int fb0, fb1;       // far_bit_counts       (125)
  unsigned nh0, nh1;  // binary_near_history  (126)
  int m0, m1;         // mean_bit_counts      (126)
  float h0, h1;       // histogram            (126)
  int bc0, bc1;       // bit_counts           (125): the last block's
};

struct Scalars {  // the wave-uniform part of AspAecDelayState (in SGPRs: every load goes through v_readfirstlane)
  int far_init, near_init;
  int minimum_probability, last_delay_probability, last_delay, last_candidate_delay, compare_delay, candidate_hits;
  float last_delay_histogram;
  int lookahead, allowed_offset;
  int previous_delay, delay_correction_count, shift_offset;
  float delay_quality_threshold;
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// lane i takes lane i - 1's value (lane 0: not used by the callers)
#ifndef AEC_DELAY_DPP_SHIFT
#define AEC_DELAY_DPP_SHIFT 1  // v_mov_b32_dpp wave_shr:1 (0: ds_bpermute)
#endif
__device__ __forceinline__ int shr1(int v, int lane) {
#if AEC_DELAY_DPP_SHIFT
  (void)lane;
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
#else
  return __builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, v);
#endif
}
// entry idx (wave-uniform, 0..127) of an array held as (v0, v1)
__device__ __forceinline__ int pick(int v0, int v1, int idx) {
  const int a = __builtin_amdgcn_readlane(v0, idx & 63), b = __builtin_amdgcn_readlane(v1, idx & 63);
  return idx < 64 ? a : b;
}
__device__ __forceinline__ float pickf(float v0, float v1, int idx) {
  return __int_as_float(pick(__float_as_int(v0), __float_as_int(v1), idx));
}
// reductions over the wave: four DPP steps inside each row of 16, then the four rows through SGPRs
#define ASP_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ int wave_min_i(int v) {
  int o;
  o = ASP_DPP(v, 0xB1); v = o < v ? o : v;   // quad_perm [1, 0, 3, 2]
  o = ASP_DPP(v, 0x4E); v = o < v ? o : v;   // quad_perm [2, 3, 0, 1]
  o = ASP_DPP(v, 0x141); v = o < v ? o : v;  // row_half_mirror
  o = ASP_DPP(v, 0x140); v = o < v ? o : v;  // row_mirror
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const int ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
__device__ __forceinline__ int wave_max_i(int v) {
  int o;
  o = ASP_DPP(v, 0xB1); v = o > v ? o : v;
  o = ASP_DPP(v, 0x4E); v = o > v ? o : v;
  o = ASP_DPP(v, 0x141); v = o > v ? o : v;
  o = ASP_DPP(v, 0x140); v = o > v ? o : v;
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const int ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
#undef ASP_DPP

__device__ __forceinline__ void mean_fix(int new_value, int factor, int& mean_value) {  // delay_estimator.c:672-684
  int diff = new_value - mean_value;
  if (diff < 0) {
    diff = -((-diff) >> factor);
  } else {
    diff = (diff >> factor);
  }
  mean_value += diff;
}

// one block from its two binary spectra: WebRtc_AddBinaryFarSpectrum + WebRtc_ProcessBinarySpectrum; returns last_delay
__device__ __forceinline__ int estimator_block_bits(Hist& H, Scalars& sc, unsigned bfar, unsigned bnear_new, int lane) {
  const int i0 = lane, i1 = lane + 64;
  const bool has1 = i1 < kHist;
  // ---- far end (aec_core.c:1194-1195; delay_estimator.c:356-369): the histories move up by one entry
  {
    const int top_h = __builtin_amdgcn_readlane((int)H.fh0, 63), top_c = __builtin_amdgcn_readlane(H.fb0, 63);
    const int s0 = shr1((int)H.fh0, lane), s1 = shr1((int)H.fh1, lane), c0 = shr1(H.fb0, lane), c1 = shr1(H.fb1, lane);
    H.fh0 = lane == 0 ? bfar : (unsigned)s0;
    H.fb0 = lane == 0 ? __popc(bfar) : c0;
    H.fh1 = lane == 0 ? (unsigned)top_h : (unsigned)s1;
    H.fb1 = lane == 0 ? top_c : c1;
  }
  // ---- near end (:1196-1197): shift the near history, pull out the delayed spectrum
  {
    const int top = __builtin_amdgcn_readlane((int)H.nh0, 63);
    const int s0 = shr1((int)H.nh0, lane), s1 = shr1((int)H.nh1, lane);
    H.nh0 = lane == 0 ? bnear_new : (unsigned)s0;
    H.nh1 = lane == 0 ? (unsigned)top : (unsigned)s1;
  }
  const unsigned bnear = (unsigned)pick((int)H.nh0, (int)H.nh1, sc.lookahead);
  // bit counts and their smoothed version (delay_estimator.c:541-560)
  int m1 = kMaxBitCountsQ9;
  {
    const int bc0 = __popc(bnear ^ H.fh0);
    H.bc0 = bc0;
    if (H.fb0 > 0) mean_fix(bc0 << 9, 13 - ((3 * H.fb0) >> 4), H.m0);
    if (has1) {
      const int bc1 = __popc(bnear ^ H.fh1);
      H.bc1 = bc1;
      if (H.fb1 > 0) mean_fix(bc1 << 9, 13 - ((3 * H.fb1) >> 4), H.m1);
      m1 = H.m1;
    }
  }
  const int m0 = H.m0;
  // best (first minimum below 32 in Q9) and worst candidates (:564-574): value * 128 + index orders by value, then index
  int key = kMaxBitCountsQ9 * 128 + 127;
  if (m0 < kMaxBitCountsQ9) key = m0 * 128 + i0;
  if (has1 && m1 < kMaxBitCountsQ9 && m1 * 128 + i1 < key) key = m1 * 128 + i1;
  key = wave_min_i(key);
  int worst = m0 > 0 ? m0 : 0;
  if (has1 && m1 > worst) worst = m1;
  worst = wave_max_i(worst);
  const int value_best_candidate = key >> 7;
  const int candidate_delay = (key & 127) == 127 ? -1 : (key & 127);
  const int valley_depth = worst - value_best_candidate;
  if ((sc.minimum_probability > 8704) && (valley_depth > 2816)) {  // kProbabilityLowerLimit, kProbabilityMinSpread
    int threshold = value_best_candidate + 1024;                     // kProbabilityOffset
    if (threshold < 8704) threshold = 8704;
    if (sc.minimum_probability > threshold) sc.minimum_probability = threshold;
  }
  sc.last_delay_probability++;
  int valid_candidate = (valley_depth > 1024) && ((value_best_candidate < sc.minimum_probability) ||
                                                  (value_best_candidate < sc.last_delay_probability));
  if (candidate_delay >= 0) {  // (-1 needs every smoothed count at 32: not reachable from the initial 20)
    // ---- UpdateRobustValidationStatistics (:90-146)
    const float kQ14Scaling = 1.f / (1 << 14);
    const float valley = valley_depth * kQ14Scaling;
    float decrease_in_last_set = valley;
    const int max_hits_for_slow_change = (candidate_delay < sc.last_delay) ? 10 : 1000;
    if (candidate_delay != sc.last_candidate_delay) {
      sc.candidate_hits = 0;
      sc.last_candidate_delay = candidate_delay;
    }
    sc.candidate_hits++;
    if (sc.candidate_hits < max_hits_for_slow_change)
      decrease_in_last_set = (pick(H.m0, H.m1, sc.compare_delay) - value_best_candidate) * kQ14Scaling;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = t == 0 ? i0 : i1;
      if (i < kHist) {
        float h = t == 0 ? H.h0 : H.h1;
        if (i == candidate_delay) {
          h += valley;
          if (h > 3000.f) h = 3000.f;
        }
        const int is_in_last_set = (i >= sc.last_delay - 2) && (i <= sc.last_delay + 1) && (i != candidate_delay);
        const int is_in_candidate_set = (i >= candidate_delay - 2) && (i <= candidate_delay + 1);
        h -= decrease_in_last_set * is_in_last_set + valley * (!is_in_last_set && !is_in_candidate_set);
        if (h < 0) h = 0;
        if (t == 0) {
          H.h0 = h;
        } else {
          H.h1 = h;
        }
      }
    }
    // ---- HistogramBasedValidation (:173-214) and RobustValidation (:236-258)
    float fraction = 1.f;
    float histogram_threshold = pickf(H.h0, H.h1, sc.compare_delay);
    const int delay_difference = candidate_delay - sc.last_delay;
    if (delay_difference > sc.allowed_offset) {
      fraction = 1.f - 0.05f * (delay_difference - sc.allowed_offset);
      fraction = (fraction > 0.5f ? fraction : 0.5f);
    } else if (delay_difference < 0) {
      fraction = 0.25f - 0.05f * delay_difference;
      fraction = (fraction > 1.f ? 1.f : fraction);
    }
    histogram_threshold *= fraction;
    histogram_threshold = (histogram_threshold > 1.5f ? histogram_threshold : 1.5f);
    const float h_cand = pickf(H.h0, H.h1, candidate_delay);
    const int is_histogram_valid = (h_cand >= histogram_threshold) && (sc.candidate_hits > 10);
    int is_robust = (sc.last_delay < 0) && (valid_candidate || is_histogram_valid);
    is_robust |= valid_candidate && is_histogram_valid;
    is_robust |= is_histogram_valid && (h_cand > sc.last_delay_histogram);
    valid_candidate = is_robust;
    if (valid_candidate) {  // :619-641
      if (candidate_delay != sc.last_delay) {
        sc.last_delay_histogram = (h_cand > 250.f ? 250.f : h_cand);
        const float h_cmp = pickf(H.h0, H.h1, sc.compare_delay);
        if (h_cand < h_cmp) {
          if (i0 == sc.compare_delay) H.h0 = h_cand;
          if (i1 == sc.compare_delay) H.h1 = h_cand;
        }
      }
      sc.last_delay = candidate_delay;
      if (value_best_candidate < sc.last_delay_probability) sc.last_delay_probability = value_best_candidate;
      sc.compare_delay = sc.last_delay;
    }
  }
  return sc.last_delay;
}

// one block: AddFarSpectrum + DelayEstimatorProcessFloat; returns last_delay
__device__ __forceinline__ int estimator_block(Hist& H, Scalars& sc, float far_pow, float near_pow, float& thr_far,
                                               float& thr_near, int lane) {
  const unsigned bfar = binary_spectrum(sqrtf(far_pow), thr_far, sc.far_init, lane);
  const unsigned bnear = binary_spectrum(sqrtf(near_pow), thr_near, sc.near_init, lane);
  return estimator_block_bits(H, sc, bfar, bnear, lane);
}

// WebRtc_SoftResetDelayEstimator + ...Farend (delay_estimator.c:500-511, 309-339) by `delay_shift` partitions
__device__ __forceinline__ void soft_reset(Hist& H, Scalars& sc, int delay_shift, int lane) {
  sc.lookahead -= delay_shift;
  if (sc.lookahead < 0) sc.lookahead = 0;
  if (sc.lookahead > kNearHist - 1) sc.lookahead = kNearHist - 1;
  if (delay_shift == 0) return;
  // entry i takes entry i - delay_shift (zero outside the history)
  unsigned h[2];
  int c[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = lane + 64 * t, src = i - delay_shift;
    const bool ok = i < kHist && src >= 0 && src < kHist;
    const int addr = (src & 63) << 2;
    const int ha = __builtin_amdgcn_ds_bpermute(addr, (int)H.fh0), hb = __builtin_amdgcn_ds_bpermute(addr, (int)H.fh1);
    const int ca = __builtin_amdgcn_ds_bpermute(addr, H.fb0), cb = __builtin_amdgcn_ds_bpermute(addr, H.fb1);
    h[t] = ok ? (unsigned)(src < 64 ? ha : hb) : 0u;
    c[t] = ok ? (src < 64 ? ca : cb) : 0;
  }
  H.fh0 = h[0];
  H.fh1 = h[1];
  H.fb0 = c[0];
  H.fb1 = c[1];
}

// SC1: the hand-off build -- a stream's estimator block passes from one frame step's wave to the next without a
// launch boundary, so every access bypasses the L1 / writes through (agent scope), like the state block
template <bool SC1, class T>
__device__ __forceinline__ T est_ld(const T* p) {
  typedef __attribute__((address_space(1))) T gT;
  if constexpr (SC1) return __hip_atomic_load((const gT*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool SC1, class T, class V>
__device__ __forceinline__ void est_st(T* p, V v) {
  typedef __attribute__((address_space(1))) T gT;
  if constexpr (SC1) __hip_atomic_store((gT*)p, (T)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = (T)v;
}

// the arrays and the scalars into registers
template <bool SC1 = false>
__device__ __forceinline__ void load_estimator(const AspAecDelayState* __restrict__ g, Hist& H, Scalars& sc, int lane) {
  const int i0 = lane, i1 = lane + 64;
  H.fh0 = est_ld<SC1>(&g->binary_far_history[i0]);
  H.fb0 = est_ld<SC1>(&g->far_bit_counts[i0]);
  H.nh0 = est_ld<SC1>(&g->binary_near_history[i0]);
  H.m0 = est_ld<SC1>(&g->mean_bit_counts[i0]);
  H.h0 = est_ld<SC1>(&g->histogram[i0]);
  H.bc0 = est_ld<SC1>(&g->bit_counts[i0]);
  H.fh1 = i1 < kHist ? est_ld<SC1>(&g->binary_far_history[i1]) : 0u;
  H.fb1 = i1 < kHist ? est_ld<SC1>(&g->far_bit_counts[i1]) : 0;
  H.bc1 = i1 < kHist ? est_ld<SC1>(&g->bit_counts[i1]) : 0;
  H.nh1 = i1 < kNearHist ? est_ld<SC1>(&g->binary_near_history[i1]) : 0u;
  H.m1 = i1 < kHist + 1 ? est_ld<SC1>(&g->mean_bit_counts[i1]) : 0;
  H.h1 = i1 < kHist + 1 ? est_ld<SC1>(&g->histogram[i1]) : 0.f;
  sc.far_init = uni(est_ld<SC1>(&g->far_spectrum_initialized));
  sc.near_init = uni(est_ld<SC1>(&g->near_spectrum_initialized));
  sc.minimum_probability = uni(est_ld<SC1>(&g->minimum_probability));
  sc.last_delay_probability = uni(est_ld<SC1>(&g->last_delay_probability));
  sc.last_delay = uni(est_ld<SC1>(&g->last_delay));
  sc.last_candidate_delay = uni(est_ld<SC1>(&g->last_candidate_delay));
  sc.compare_delay = uni(est_ld<SC1>(&g->compare_delay));
  sc.candidate_hits = uni(est_ld<SC1>(&g->candidate_hits));
  sc.last_delay_histogram = unif(est_ld<SC1>(&g->last_delay_histogram));
  sc.lookahead = uni(est_ld<SC1>(&g->lookahead));
  sc.allowed_offset = uni(est_ld<SC1>(&g->allowed_offset));
  sc.previous_delay = uni(est_ld<SC1>(&g->previous_delay));
  sc.delay_correction_count = uni(est_ld<SC1>(&g->delay_correction_count));
  sc.shift_offset = uni(est_ld<SC1>(&g->shift_offset));
  sc.delay_quality_threshold = unif(est_ld<SC1>(&g->delay_quality_threshold));
}

// back to HBM.  kSpectra: with the mean spectra's initialised flags (the hand-off build of the process kernel keeps
// those and the mean spectra themselves: aec_kernels.hip, flow_binary_spectra)
template <bool kSpectra, bool SC1 = false>
__device__ __forceinline__ void store_estimator(AspAecDelayState* __restrict__ g, const Hist& H, const Scalars& sc, int lane) {
  const int i0 = lane, i1 = lane + 64;
  est_st<SC1>(&g->binary_far_history[i0], H.fh0);
  est_st<SC1>(&g->far_bit_counts[i0], H.fb0);
  est_st<SC1>(&g->binary_near_history[i0], H.nh0);
  est_st<SC1>(&g->mean_bit_counts[i0], H.m0);
  est_st<SC1>(&g->histogram[i0], H.h0);
  est_st<SC1>(&g->bit_counts[i0], H.bc0);
  if (i1 < kHist) {
    est_st<SC1>(&g->binary_far_history[i1], H.fh1);
    est_st<SC1>(&g->far_bit_counts[i1], H.fb1);
    est_st<SC1>(&g->bit_counts[i1], H.bc1);
  }
  if (i1 < kNearHist) est_st<SC1>(&g->binary_near_history[i1], H.nh1);
  if (i1 < kHist + 1) {
    est_st<SC1>(&g->mean_bit_counts[i1], H.m1);
    est_st<SC1>(&g->histogram[i1], H.h1);
  }
  if (lane == 0) {
    if (kSpectra) {
      est_st<SC1>(&g->far_spectrum_initialized, sc.far_init);
      est_st<SC1>(&g->near_spectrum_initialized, sc.near_init);
    }
    est_st<SC1>(&g->minimum_probability, sc.minimum_probability);
    est_st<SC1>(&g->last_delay_probability, sc.last_delay_probability);
    est_st<SC1>(&g->last_delay, sc.last_delay);
    est_st<SC1>(&g->last_candidate_delay, sc.last_candidate_delay);
    est_st<SC1>(&g->compare_delay, sc.compare_delay);
    est_st<SC1>(&g->candidate_hits, sc.candidate_hits);
    est_st<SC1>(&g->last_delay_histogram, sc.last_delay_histogram);
    est_st<SC1>(&g->lookahead, sc.lookahead);
    est_st<SC1>(&g->previous_delay, sc.previous_delay);
    est_st<SC1>(&g->delay_correction_count, sc.delay_correction_count);
    est_st<SC1>(&g->shift_offset, sc.shift_offset);
    est_st<SC1>(&g->delay_quality_threshold, sc.delay_quality_threshold);
  }
}

// The stream's far-buffer control of the coming sub-frame in the delay-agnostic mode (aec_core.c:1696-1751; ops.control
// != 0): the WebRtcAec_BufferFarend calls since the last step replayed on the stream's own read side, the under-run
// stuffing, SignalBasedDelayCorrection, the read-pointer move with the estimator's soft reset, and the far slots of
// the sub-frame's blocks (also left in blk->slot for a process launch that reads them from memory).
template <bool SC1 = false, class OPS>
__device__ __forceinline__ void control_step(DelayBlock* __restrict__ blk, Hist& H, Scalars& sc, const OPS& ops, int lane,
                                             int& slot0_out, int& slot1_out) {
  AspAecDelayState* g = &blk->s;
  FarPos fp;
  int sd;
  if (ops.sync) {
    fp.read = ops.h_far_read;
    fp.write = ops.h_far_write;
    fp.wrap = ops.h_far_wrap;
    sd = ops.h_system_delay;
  } else {
    fp.read = uni(est_ld<SC1>(&g->far_read));
    fp.write = uni(est_ld<SC1>(&g->far_write));
    fp.wrap = uni(est_ld<SC1>(&g->far_wrap));
    sd = uni(est_ld<SC1>(&g->system_delay));
  }
  for (int e = 0; e < ops.nevents; ++e) {  // WebRtcAec_BufferFarend since the last step (echo_cancellation.c:316-336)
    sd += ops.ev_samples[e];
    for (int p = 0; p < ops.ev_parts[e]; ++p) {
      if (kFarSlots - fp_avail_read(fp) < 1) sd -= fp_move_read(fp, 1) * kPartLen;  // aec_core.c:1622-1625
      fp_write_one(fp);
    }
  }
  int slot0 = 0, slot1 = 0;
  if (ops.control == 1) {  // (2: only the replay above)
  if (sd < kFrameLen) sd -= fp_move_read(fp, -(ops.mult + 1)) * kPartLen;  // 1) aec_core.c:1696-1700
  {
    // SignalBasedDelayCorrection (aec_core.c:797-850)
    const float quality = pickf(H.h0, H.h1, sc.compare_delay) / 3000.f;  // WebRtc_binary_last_delay_quality, robust validation on
    int delay_correction = 0;
    const int last_delay = sc.last_delay;
    if ((last_delay >= 0) && (last_delay != sc.previous_delay) && (quality > sc.delay_quality_threshold)) {
      const int delay = last_delay - sc.lookahead;
      if (delay <= 0 || delay > (ops.num_part / 4)) {
        const int available_read = fp_avail_read(fp);
        delay_correction = -(delay - sc.shift_offset);
        sc.shift_offset--;
        sc.shift_offset = (sc.shift_offset <= 1 ? 1 : sc.shift_offset);
        if (delay_correction > available_read - ops.mult - 1) {
          delay_correction = 0;
        } else {
          sc.previous_delay = last_delay;
          ++sc.delay_correction_count;
        }
      }
    }
    if (sc.delay_correction_count > 0) {
      float delay_quality = quality;
      delay_quality = (delay_quality > 0.07f ? 0.07f : delay_quality);  // kDelayQualityThresholdMax
      sc.delay_quality_threshold = (delay_quality > sc.delay_quality_threshold ? delay_quality : sc.delay_quality_threshold);
    }
    const int moved_elements = fp_move_read(fp, delay_correction);  // 2 b) aec_core.c:1719-1730
    soft_reset(H, sc, moved_elements, lane);
    if (fp_avail_read(fp) < (ops.mult + 1)) sd -= fp_move_read(fp, -(ops.mult + 1)) * kPartLen;  // :1747-1750
  }
  for (int k = 0; k < ops.nblocks; ++k) {  // WebRtc_ReadBuffer(far_buf) of each block to come (aec_core.c:1140-1141)
    const int s = fp.read >= kFarSlots ? fp.read - kFarSlots : fp.read;
    if (k == 0) slot0 = s; else slot1 = s;
    fp_move_read(fp, 1);
  }
  sd -= kFrameLen;  // 5) aec_core.c:1758
  }
  if (lane == 0) {
    est_st<SC1>(&g->far_read, fp.read);
    est_st<SC1>(&g->far_write, fp.write);
    est_st<SC1>(&g->far_wrap, fp.wrap);
    est_st<SC1>(&g->system_delay, sd);
    est_st<SC1>(&blk->slot[0], slot0);
    est_st<SC1>(&blk->slot[1], slot1);
  }
  slot0_out = slot0;
  slot1_out = slot1;
}

}  // namespace aspaec_est
