// bt_sure.h -- the SURE pass of bt_kernels8.hip (lane = macro-column, the two half-waves take the two halves of a
// segmentation's blocks, packed lean divisions) for the run-time-sized kernel of bt_kernels.hip: the same code with
// the squared-real table's row stride as a template parameter and the column count at run time.  (bt_kernels8.hip
// keeps its own copy: its register allocation is tuned around it.)  At most 32 macro-columns per call: lane & 31.
#pragma once
#include <hip/hip_runtime.h>

#include "bt_layout.h"
#include "pk_f32.h"

namespace aspbt_sure {
using namespace aspbt;
using asppk::f32x2;
constexpr int SUS = 15;  // SURE values per macro-column

// Correctly rounded n / d by the Newton + residual steps hipcc emits for `/`, without its range scaling;
// exact while d and n / d are normal floats far from the range limits (callers check and fall back).
__device__ __forceinline__ float fdiv_lean(float n, float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float q0 = n * r1;
  const float e1 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ float fdiv_checked(float n, float d) {
  float q = fdiv_lean(n, d);
  if (__builtin_expect(!(d >= 1e-18f && d <= 1e18f), 0)) q = n / d;
  return q;
}

__device__ __forceinline__ f32x2 fdiv2_lean(f32x2 n, f32x2 d) {  // two quotients, packed-f32 issue slots
  f32x2 r0;
  r0.x = __builtin_amdgcn_rcpf(d.x);
  r0.y = __builtin_amdgcn_rcpf(d.y);
  const f32x2 one = {1.0f, 1.0f};
  const f32x2 e0 = __builtin_elementwise_fma(-d, r0, one);
  const f32x2 r1 = __builtin_elementwise_fma(e0, r0, r0);
  const f32x2 q0 = n * r1;
  const f32x2 e1 = __builtin_elementwise_fma(-d, q0, n);
  const f32x2 q1 = __builtin_elementwise_fma(e1, r1, q0);
  const f32x2 e2 = __builtin_elementwise_fma(-d, q1, n);
  return __builtin_elementwise_fma(e2, r1, q1);
}

// one term of the SURE sum (.c:391-398); q = temp / e
__device__ __forceinline__ float sure_term_q(float e, float q, float size_blk, float thr, float two_size) {
  return size_blk + q * (float)(e > thr) + (e - two_size) * (float)(e <= thr);
}
// The terms of NT block energies, in place (x: energies in, terms out): lean divisions, two terms per
// packed instruction, when every energy is inside the range the lean division is exact for; IEEE
// divisions otherwise (rare: one branch per segmentation).
template <int NT>
__device__ __forceinline__ void sure_terms(float (&x)[NT], float temp, float size_blk, float thr,
                                           float two_size) {
  float emin = x[0], emax = x[0];
  if constexpr (NT > 1) {
    emin = fminf(x[0], x[1]);
    emax = fmaxf(x[0], x[1]);
#pragma unroll
    for (int q = 2; q < NT; q += 2) {
      emin = __builtin_fminf(emin, __builtin_fminf(x[q], x[q + 1]));  // v_min3_f32
      emax = __builtin_fmaxf(emax, __builtin_fmaxf(x[q], x[q + 1]));
    }
  }
  if (__builtin_expect(emin >= 1e-18f && emax <= 1e18f, 1)) {
    if constexpr (NT == 1) {
      x[0] = sure_term_q(x[0], fdiv_lean(temp, x[0]), size_blk, thr, two_size);
    } else {
      const f32x2 temp2 = {temp, temp}, size2 = {size_blk, size_blk}, two2 = {two_size, two_size};
#pragma unroll
      for (int q = 0; q < NT; q += 2) {
        const f32x2 e2 = {x[q], x[q + 1]};
        const f32x2 q2 = fdiv2_lean(temp2, e2);
        const f32x2 g2 = {(float)(e2.x > thr), (float)(e2.y > thr)};
        const f32x2 l2 = f32x2{1.0f, 1.0f} - g2;  // (float)(e <= thr) for every e that is not a NaN (a NaN term is a NaN either way)
        const f32x2 t2 = size2 + q2 * g2 + (e2 - two2) * l2;
        x[q] = t2.x;
        x[q + 1] = t2.y;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < NT; ++q) x[q] = sure_term_q(x[q], temp / x[q], size_blk, thr, two_size);
  }
}

// Sequential (rows outer, columns inner) sum of one TT x FF block of the squared-real table: compile-time
// offsets, so the LDS reads pipeline while the adds keep the reference's order (.c:322-338).
template <int TT, int FF, int SQS>
__device__ __forceinline__ float block_sum(const float* col, int r0, int c0) {
  float acc = 0.0f;
#pragma unroll
  for (int r = 0; r < TT; ++r)
#pragma unroll
    for (int c = 0; c < FF; ++c) acc += col[((r0 + r) * 16 + (c0 + c)) * SQS];
  return acc;
}

// SURE of segmentation (T, F) for the macro-columns of a wave's lanes (.c:378-400).  Half-wave h takes
// the blocks of the second half of the (ii major, jj minor) order when h = 1: rows 4..7 for T >= 1,
// columns 8..15 for T = 0; its running sum starts from the first half's total.
template <int T, int F, int SQS>
__device__ __forceinline__ void sure_seg(const float* sq, float* sure, const BtSeg* __restrict__ sgp, int lane, int ncol) {
  const float temp = sgp->temp, size_blk = sgp->size_blk, thr = sgp->thr, two_size = sgp->two_size;
  constexpr int TT = 8 >> T, FF = 16 >> F, S = T + F, c = T * 5 + F;
  const int h = lane >> 5, m = lane & 31;
  if constexpr (S == 0) {
    float t[1];
    t[0] = block_sum<8, 16, SQS>(sq + m, 0, 0);
    sure_terms<1>(t, temp, size_blk, thr, two_size);
    float s = 0.0f;
    s += t[0];
    if (lane < ncol) sure[m * SUS + c] = s;
  } else {
    constexpr int NTERM = 1 << (S - 1);
    constexpr int NJ = T >= 1 ? (1 << F) : (1 << (F - 1));  // jj values per half
    const float* col = sq + m + (T >= 1 ? h * (4 * 16 * SQS) : h * (8 * SQS));
    // terms in chunks of at most 8 blocks (a scheduling fence between chunks keeps the LDS reads of later
    // chunks from being hoisted over the whole segmentation: 128 VGPRs, four waves per SIMD)
    constexpr int CH = NTERM < 8 ? NTERM : 8;
    float t[NTERM];
#pragma unroll
    for (int c0 = 0; c0 < NTERM; c0 += CH) {
      float x[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) x[q] = block_sum<TT, FF, SQS>(col, TT * ((c0 + q) / NJ), FF * ((c0 + q) % NJ));
      sure_terms<CH>(x, temp, size_blk, thr, two_size);
#pragma unroll
      for (int q = 0; q < CH; ++q) t[c0 + q] = x[q];
      __builtin_amdgcn_sched_barrier(0);
    }
    float s0 = 0.0f;
#pragma unroll
    for (int q = 0; q < NTERM; ++q) s0 += t[q];
    float s1 = __shfl_xor(s0, 32);  // the first half's total, seen from the second half
#pragma unroll
    for (int q = 0; q < NTERM; ++q) s1 += t[q];
    if (h == 1 && m < ncol) sure[m * SUS + c] = s1;
  }
}

// the fifteen segmentations dealt to eight slots by cost (as in bt_kernels8.hip); a workgroup of four waves runs
// slots w and w + 4
template <int SQS>
__device__ __forceinline__ void sure_slot(int slot, const float* sq, float* sure, const BtSize& P, int lane, int ncol) {
  switch (slot) {
    case 0: sure_seg<2, 4, SQS>(sq, sure, &P.seg[2][4], lane, ncol); break;
    case 1: sure_seg<2, 3, SQS>(sq, sure, &P.seg[2][3], lane, ncol); sure_seg<0, 1, SQS>(sq, sure, &P.seg[0][1], lane, ncol); break;
    case 2: sure_seg<1, 4, SQS>(sq, sure, &P.seg[1][4], lane, ncol); sure_seg<1, 0, SQS>(sq, sure, &P.seg[1][0], lane, ncol); break;
    case 3: sure_seg<2, 2, SQS>(sq, sure, &P.seg[2][2], lane, ncol); sure_seg<0, 0, SQS>(sq, sure, &P.seg[0][0], lane, ncol); break;
    case 4: sure_seg<1, 3, SQS>(sq, sure, &P.seg[1][3], lane, ncol); sure_seg<2, 1, SQS>(sq, sure, &P.seg[2][1], lane, ncol); break;
    case 5: sure_seg<0, 4, SQS>(sq, sure, &P.seg[0][4], lane, ncol); sure_seg<1, 2, SQS>(sq, sure, &P.seg[1][2], lane, ncol); break;
    case 6: sure_seg<0, 3, SQS>(sq, sure, &P.seg[0][3], lane, ncol); sure_seg<0, 2, SQS>(sq, sure, &P.seg[0][2], lane, ncol); break;
    default: sure_seg<2, 0, SQS>(sq, sure, &P.seg[2][0], lane, ncol); sure_seg<1, 1, SQS>(sq, sure, &P.seg[1][1], lane, ncol); break;
  }
}

}  // namespace aspbt_sure
