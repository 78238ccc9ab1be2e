// ns_device.h -- device-side helpers shared by the NS kernels (ns_kernels.hip: one stream per
// wave, bins q / q + 64; ns_kernels1.hip: one stream per wave, pair layout): the reference's constants, exact division /
// sqrt / log / exp / tanh forms (each verified exhaustively against its libm form on the device,
// tests/test_ns_gpu.py), wave reductions, and the histogram-window close.
#pragma once
#include <hip/hip_runtime.h>

#include "ns_layout.h"

namespace aspns_dev {
using namespace aspns;

// instruction-budget builds (tools/ns_valu_budget.py, never shipped) assert that the rarely taken
// libm fallbacks are not taken, so that they vanish from the straight-line code being counted
#ifdef NS1_BUDGET
#define ASP_NS_RARE(c) (__builtin_assume(!(c)), false)
#else
#define ASP_NS_RARE(c) __builtin_expect((c), 0)
#endif

// ns/defines.h:19-48, same (float)<double literal> spelling as the reference
#define NS_QUANTILE (float)0.25
#define NS_END_STARTUP_LONG 200
#define NS_END_STARTUP_SHORT 50
#define NS_FACTOR (float)40.0
#define NS_WIDTH (float)0.01
#define NS_DD_PR_SNR (float)0.98
#define NS_LRT_TAVG (float)0.50
#define NS_SPECT_FL_TAVG (float)0.30
#define NS_SPECT_DIFF_TAVG (float)0.30
#define NS_PRIOR_UPDATE (float)0.10
#define NS_NOISE_UPDATE (float)0.90
#define NS_SPEECH_UPDATE (float)0.99
#define NS_WIDTH_PR_MAP (float)4.0
#define NS_PROB_RANGE (float)0.20
#define NS_GAMMA_PAUSE (float)0.05
#define NS_B_LIM (float)0.5
#define NS_START_BAND 5

__device__ __forceinline__ float xorf(float x, uint32_t m) {
  return __uint_as_float(__float_as_uint(x) ^ m);
}

// Orders the wave's own LDS traffic for the compiler; the hardware executes
// one wave's DS instructions in order, so no s_barrier is needed.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {  // lane: compile-time constant
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Wave64 all-reduce with the association of an ascending xor-butterfly
// (xor 1, 2, 4, 8, 16, 32; every lane ends with the same value), which
// oracle/ns_oracle.c reproduces in ASP_NS_REDUCE_TREE mode.  Steps 1 and 2 are
// DPP quad permutes; after them a quad is uniform, so the half-row and row
// mirrors deliver exactly the xor-4 / xor-8 partners' values; the four uniform
// row sums are then combined through scalar broadcasts as (r0+r1)+(r2+r3).
__device__ __forceinline__ float wave_sum(float v) {
  v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]  == xor 1
  v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]  == xor 2
  v = v + dpp_move<0x141>(v);  // row_half_mirror      == xor 4 (quads uniform)
  v = v + dpp_move<0x140>(v);  // row_mirror           == xor 8 (octets uniform)
  const float r0 = lane_bcast(v, 0), r1 = lane_bcast(v, 16);
  const float r2 = lane_bcast(v, 32), r3 = lane_bcast(v, 48);
  return (r0 + r1) + (r2 + r3);
}

// The same sum (same operands, same association: bit-identical) with the four row sums combined by two
// row-broadcast DPP adds instead of four v_readlane, two moves and two adds: row_bcast:15 (rows 1, 3
// written) leaves r0 + r1 and r2 + r3 in rows 1 and 3, row_bcast:31 (row 3 written) adds row 1's into
// row 3's: (r0 + r1) + (r2 + r3), read from lane 63.  Seven 2.8-cycle issue slots instead of eight plus
// five 1.7-cycle ones (profiles/r03_issue_probe3.txt).  A DPP read needs two wait states behind the VALU
// write of its source: one sum alone pays them as s_nops, N sums reduced side by side (wave_sums_bcast)
// fill each other's slots.
#define ASP_DPP_STEP(r, ctl) "v_add_f32_dpp " r ", " r ", " r " " ctl "\n\t"
#define ASP_DPP_Q1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define ASP_DPP_Q2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
#define ASP_DPP_HM "row_half_mirror row_mask:0xf bank_mask:0xf"
#define ASP_DPP_RM "row_mirror row_mask:0xf bank_mask:0xf"
#define ASP_DPP_B15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define ASP_DPP_B31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
__device__ __forceinline__ float wave_sum_bcast(float v) {
  asm volatile("s_nop 1\n\t" ASP_DPP_STEP("%0", ASP_DPP_Q1) "s_nop 1\n\t" ASP_DPP_STEP("%0", ASP_DPP_Q2)
               "s_nop 1\n\t" ASP_DPP_STEP("%0", ASP_DPP_HM) "s_nop 1\n\t" ASP_DPP_STEP("%0", ASP_DPP_RM)
               "s_nop 1\n\t" ASP_DPP_STEP("%0", ASP_DPP_B15) "s_nop 1\n\t" ASP_DPP_STEP("%0", ASP_DPP_B31)
               "s_nop 0"
               : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// two / three / four independent sums, each with exactly wave_sum's association
__device__ __forceinline__ void wave_sums_bcast(float& a, float& b) {
#define ASP_S2(ctl) ASP_DPP_STEP("%0", ctl) ASP_DPP_STEP("%1", ctl) "s_nop 0\n\t"
  asm volatile("s_nop 1\n\t" ASP_S2(ASP_DPP_Q1) ASP_S2(ASP_DPP_Q2) ASP_S2(ASP_DPP_HM) ASP_S2(ASP_DPP_RM)
               ASP_S2(ASP_DPP_B15) ASP_S2(ASP_DPP_B31)
               : "+v"(a), "+v"(b));
#undef ASP_S2
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}
__device__ __forceinline__ void wave_sums_bcast(float& a, float& b, float& c) {
#define ASP_S3(ctl) ASP_DPP_STEP("%0", ctl) ASP_DPP_STEP("%1", ctl) ASP_DPP_STEP("%2", ctl)
  asm volatile("s_nop 1\n\t" ASP_S3(ASP_DPP_Q1) ASP_S3(ASP_DPP_Q2) ASP_S3(ASP_DPP_HM) ASP_S3(ASP_DPP_RM)
               ASP_S3(ASP_DPP_B15) ASP_S3(ASP_DPP_B31) "s_nop 0"
               : "+v"(a), "+v"(b), "+v"(c));
#undef ASP_S3
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
}
__device__ __forceinline__ void wave_sums_bcast(float& a, float& b, float& c, float& d) {
#define ASP_S4(ctl) ASP_DPP_STEP("%0", ctl) ASP_DPP_STEP("%1", ctl) ASP_DPP_STEP("%2", ctl) ASP_DPP_STEP("%3", ctl)
  asm volatile("s_nop 1\n\t" ASP_S4(ASP_DPP_Q1) ASP_S4(ASP_DPP_Q2) ASP_S4(ASP_DPP_HM) ASP_S4(ASP_DPP_RM)
               ASP_S4(ASP_DPP_B15) ASP_S4(ASP_DPP_B31) "s_nop 0"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef ASP_S4
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
  d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
}

// Correctly rounded a / d for a divisor shared by the whole wave: `rd` is the
// correctly rounded reciprocal of d (one exact division per wave), then
// Markstein's q0 = a*rd, r = a - d*q0, q = q0 + r*rd is the rounded quotient
// (needs no range scaling here: |a/d| and d stay far from the float limits).
__device__ __forceinline__ float div_by_uniform(float a, float d, float rd) {
  const float q0 = a * rd;
  const float r = __builtin_fmaf(-d, q0, a);
  return __builtin_fmaf(r, rd, q0);
}

// Correctly rounded n / d without the range scaling / special-case fix-up of
// the generic expansion: the same Newton + residual arithmetic hipcc emits for
// `/` (v_rcp, two reciprocal refinements, two quotient corrections), valid
// while d and n/d are normal floats away from the range limits -- true for
// every use below (denominators are x + 1e-4, 1 + x, overdrive + x or a
// density > 0 with int16-range audio).  Checked against `/` on the device over
// full mantissa sweeps (tests/test_ns_gpu.py).
__device__ __forceinline__ float fdiv(float n, float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float q0 = n * r1;
  const float e1 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e2, r1, q1);
}
// Two correctly rounded divisions at a time: the same arithmetic as fdiv with the seven
// multiply-add steps issued as packed-f32 instructions (one issue slot for two quotients).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fdiv2(f32x2 n, f32x2 d) {
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
// Three bins of a lane of the one-stream-per-wave kernel (two owned + bin 128) as one packed pair
// and a scalar: +, -, * and fma on an F3 compile to v_pk_* for the pair.  Every operation is the IEEE
// single operation of its scalar spelling (no contraction), so results are bit-identical to per-bin loops.
struct B3 {
  bool v[3];
};
struct F3 {
  f32x2 p;
  float t;
  __device__ __forceinline__ F3() {}
  __device__ __forceinline__ F3(f32x2 p_, float t_) : p(p_), t(t_) {}
  __device__ __forceinline__ explicit F3(float c) : p(f32x2{c, c}), t(c) {}
  __device__ __forceinline__ explicit F3(const float (&x)[3]) : p(f32x2{x[0], x[1]}), t(x[2]) {}
  __device__ __forceinline__ void store(float (&x)[3]) const {
    x[0] = p.x; x[1] = p.y; x[2] = t;
  }
};
__device__ __forceinline__ F3 operator+(const F3& x, const F3& y) { return F3(x.p + y.p, x.t + y.t); }
__device__ __forceinline__ F3 operator-(const F3& x, const F3& y) { return F3(x.p - y.p, x.t - y.t); }
__device__ __forceinline__ F3 operator*(const F3& x, const F3& y) { return F3(x.p * y.p, x.t * y.t); }
__device__ __forceinline__ F3 operator*(float c, const F3& y) { return F3(c) * y; }
__device__ __forceinline__ F3 operator*(const F3& x, float c) { return x * F3(c); }
__device__ __forceinline__ F3 operator+(const F3& x, float c) { return x + F3(c); }
__device__ __forceinline__ F3 operator-(const F3& x, float c) { return x - F3(c); }
__device__ __forceinline__ F3 operator-(float c, const F3& y) { return F3(c) - y; }
__device__ __forceinline__ F3 fma3(const F3& x, const F3& y, const F3& z) {
  return F3(__builtin_elementwise_fma(x.p, y.p, z.p), __builtin_fmaf(x.t, y.t, z.t));
}
// One v_max_f32 / v_min_f32 (the builtin forms put a canonicalising v_max x, x in front of each).  For a
// quiet NaN they return the other operand, as `x > c ? x : c` / `x < c ? x : c` do.
__device__ __forceinline__ float fmax_raw(float x, float c) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
  return r;
}
__device__ __forceinline__ float fmin_raw(float x, float c) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
  return r;
}
__device__ __forceinline__ F3 min3(const F3& x, const F3& y) {  // y < x ? y : x, NaN-free operands
  return F3(f32x2{fmin_raw(x.p.x, y.p.x), fmin_raw(x.p.y, y.p.y)}, fmin_raw(x.t, y.t));
}
__device__ __forceinline__ F3 max3(const F3& x, float c) {
  return F3(f32x2{fmax_raw(x.p.x, c), fmax_raw(x.p.y, c)}, fmax_raw(x.t, c));
}
__device__ __forceinline__ F3 abs3(const F3& x) {
  return F3(f32x2{fabsf(x.p.x), fabsf(x.p.y)}, fabsf(x.t));
}
#define ASP_F3_CMP(name, op)                                                        \
  __device__ __forceinline__ B3 name(const F3& x, const F3& y) {                    \
    B3 r;                                                                           \
    r.v[0] = x.p.x op y.p.x; r.v[1] = x.p.y op y.p.y; r.v[2] = x.t op y.t;          \
    return r;                                                                       \
  }
ASP_F3_CMP(gt3, >)
ASP_F3_CMP(lt3, <)
#undef ASP_F3_CMP
__device__ __forceinline__ F3 sel3(const B3& c, const F3& x, const F3& y) {  // c ? x : y
  return F3(f32x2{c.v[0] ? x.p.x : y.p.x, c.v[1] ? x.p.y : y.p.y}, c.v[2] ? x.t : y.t);
}
__device__ __forceinline__ F3 div_by_uniform3(const F3& a, float d, float rd) {
  const F3 q0 = a * rd;
  const F3 r = fma3(F3(-d), q0, a);
  return fma3(r, F3(rd), q0);
}
__device__ __forceinline__ F3 fdiv3v(const F3& n, const F3& d) { return F3(fdiv2(n.p, d.p), fdiv(n.t, d.t)); }
__device__ __forceinline__ void fdiv3(const float (&n)[3], const float (&d)[3], float (&q)[3]) {
  const f32x2 a = fdiv2(f32x2{n[0], n[1]}, f32x2{d[0], d[1]});
  q[0] = a.x; q[1] = a.y;
  q[2] = fdiv(n[2], d[2]);
}
// ---- the two-streams-per-wave kernel (ns_kernels2.hip): four owned bins + bin 128 per lane
__device__ __forceinline__ void fdiv5(const float (&n)[5], const float (&d)[5], float (&q)[5]) {
  const f32x2 a = fdiv2(f32x2{n[0], n[1]}, f32x2{d[0], d[1]});
  const f32x2 b = fdiv2(f32x2{n[2], n[3]}, f32x2{d[2], d[3]});
  q[0] = a.x; q[1] = a.y; q[2] = b.x; q[3] = b.y;
  q[4] = fdiv(n[4], d[4]);
}
// Five bins of a lane (four owned + bin 128) as two packed pairs and a scalar: +, -, * and fma on
// an F5 compile to v_pk_* for the pairs.  Every operation is the IEEE single operation of its
// scalar spelling (no contraction), so results are bit-identical to the per-bin loops.
struct B5 {
  bool v[5];
};
struct F5 {
  f32x2 a, b;
  float t;
  __device__ __forceinline__ F5() {}
  __device__ __forceinline__ F5(f32x2 a_, f32x2 b_, float t_) : a(a_), b(b_), t(t_) {}
  __device__ __forceinline__ explicit F5(float c) : a(f32x2{c, c}), b(f32x2{c, c}), t(c) {}
  __device__ __forceinline__ explicit F5(const float (&x)[5]) : a(f32x2{x[0], x[1]}), b(f32x2{x[2], x[3]}), t(x[4]) {}
  __device__ __forceinline__ void store(float (&x)[5]) const {
    x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y; x[4] = t;
  }
  __device__ __forceinline__ float get(int k) const { return k == 0 ? a.x : k == 1 ? a.y : k == 2 ? b.x : k == 3 ? b.y : t; }
};
__device__ __forceinline__ F5 operator+(const F5& x, const F5& y) { return F5(x.a + y.a, x.b + y.b, x.t + y.t); }
__device__ __forceinline__ F5 operator-(const F5& x, const F5& y) { return F5(x.a - y.a, x.b - y.b, x.t - y.t); }
__device__ __forceinline__ F5 operator*(const F5& x, const F5& y) { return F5(x.a * y.a, x.b * y.b, x.t * y.t); }
__device__ __forceinline__ F5 operator*(float c, const F5& y) { return F5(c) * y; }
__device__ __forceinline__ F5 operator*(const F5& x, float c) { return x * F5(c); }
__device__ __forceinline__ F5 operator+(const F5& x, float c) { return x + F5(c); }
__device__ __forceinline__ F5 operator-(const F5& x, float c) { return x - F5(c); }
__device__ __forceinline__ F5 operator-(float c, const F5& y) { return F5(c) - y; }
__device__ __forceinline__ F5 operator-(const F5& x) { return F5(-x.a, -x.b, -x.t); }
__device__ __forceinline__ F5 fma5(const F5& x, const F5& y, const F5& z) {
  return F5(__builtin_elementwise_fma(x.a, y.a, z.a), __builtin_elementwise_fma(x.b, y.b, z.b),
            __builtin_fmaf(x.t, y.t, z.t));
}
__device__ __forceinline__ F5 abs5(const F5& x) {
  return F5(f32x2{fabsf(x.a.x), fabsf(x.a.y)}, f32x2{fabsf(x.b.x), fabsf(x.b.y)}, fabsf(x.t));
}
#define ASP_F5_CMP(name, op)                                                        \
  __device__ __forceinline__ B5 name(const F5& x, const F5& y) {                    \
    B5 r;                                                                           \
    r.v[0] = x.a.x op y.a.x; r.v[1] = x.a.y op y.a.y; r.v[2] = x.b.x op y.b.x;      \
    r.v[3] = x.b.y op y.b.y; r.v[4] = x.t op y.t;                                   \
    return r;                                                                       \
  }
ASP_F5_CMP(gt5, >)
ASP_F5_CMP(lt5, <)
#undef ASP_F5_CMP
__device__ __forceinline__ F5 sel5(const B5& c, const F5& x, const F5& y) {  // c ? x : y
  return F5(f32x2{c.v[0] ? x.a.x : y.a.x, c.v[1] ? x.a.y : y.a.y},
            f32x2{c.v[2] ? x.b.x : y.b.x, c.v[3] ? x.b.y : y.b.y}, c.v[4] ? x.t : y.t);
}
// div_by_uniform for five bins
__device__ __forceinline__ F5 div_by_uniform5(const F5& a, float d, float rd) {
  const F5 q0 = a * rd;
  const F5 r = fma5(F5(-d), q0, a);
  return fma5(r, F5(rd), q0);
}
__device__ __forceinline__ F5 fdiv5v(const F5& n, const F5& d) {
  return F5(fdiv2(n.a, d.a), fdiv2(n.b, d.b), fdiv(n.t, d.t));
}
#define DIV129(a) div_by_uniform((a), 129.0f, 1.0f / 129.0f)

// (float)log((double)x), the reference's idiom (ns_core.c:228,540,681,1096), for
// positive finite normal x.  Lean fp64 evaluation:
//   x = 2^e m, m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1) (quotient from a float
//   reciprocal plus an exact fp64 residual); log m = 2s + s z P(z), z = s^2.
// Its error is a few 2^-52, so the float rounding is decided unless the fp64
// value sits within 2^-44 (relative) of a rounding boundary; those cases
// (about 3e-6 of inputs) and non-normal inputs are redone with the fp64 libm log,
// which makes the result identical to (float)log((double)x) from ocml for every
// float (checked exhaustively in tests/test_ns_gpu.py).
__device__ __forceinline__ double log_lean_f64(float x) {
  const int xb = __float_as_int(x);
  int e = ((xb >> 23) & 0xff) - 127;
  float mf = __int_as_float((xb & 0x007fffff) | 0x3f800000);  // [1, 2)
  const bool big = mf > 1.41421356f;
  mf = big ? mf * 0.5f : mf;  // exact
  e += big ? 1 : 0;
  const float ff = mf - 1.0f;  // exact (Sterbenz)
  const float gf = mf + 1.0f;  // rounded to float; the residual below uses g exactly in fp64
  const double g = (double)mf + 1.0;
  const double c = (double)__builtin_amdgcn_rcpf(gf);
  const double f = (double)ff;
  const double s0 = f * c;                        // exact: 24 x 24 bits
  const double d = __builtin_fma(g, c, -1.0);     // exact: g c - 1, |d| < 2^-22
  double s = __builtin_fma(-s0, d, s0);           // s0 (1 - d + d^2)
  s = __builtin_fma(s0 * d, d, s);
  const double z = s * s;
  double p = 2.0 / 21.0;
  p = __builtin_fma(p, z, 2.0 / 19.0);
  p = __builtin_fma(p, z, 2.0 / 17.0);
  p = __builtin_fma(p, z, 2.0 / 15.0);
  p = __builtin_fma(p, z, 2.0 / 13.0);
  p = __builtin_fma(p, z, 2.0 / 11.0);
  p = __builtin_fma(p, z, 2.0 / 9.0);
  p = __builtin_fma(p, z, 2.0 / 7.0);
  p = __builtin_fma(p, z, 2.0 / 5.0);
  p = __builtin_fma(p, z, 2.0 / 3.0);
  const double lm = __builtin_fma(s * z, p, s + s);
  const double ed = (double)e;
  const double ln2_hi = 0x1.62e42fefa38p-1;   // 41 significant bits: e * ln2_hi is exact
  const double ln2_lo = 0x1.ef35793c7673p-45;
  return __builtin_fma(ed, ln2_hi, __builtin_fma(ed, ln2_lo, lm));
}

__device__ __forceinline__ bool f64_rounds_safely_to_f32(double y) {
  // the 29 bits dropped by the conversion; unsafe when they are within 2^9
  // (= 2^-44 relative) of the half-way pattern 0x10000000
  const unsigned lo = (unsigned)__double2loint(y) & 0x1fffffffu;
  return ((lo - 0x0ffffe00u) > 0x400u);
}

// exp((double)x) for a float x with |x| <= 87 (float result normal), lean fp64:
// k = rint(64 x / ln 2), r = x - k ln2/64 (|r| <= 0.0055), e^r by a degree-5
// polynomial, 2^(k/64) from a 64-entry table.  Relative error < 2^-51.
__device__ __forceinline__ double exp_lean_f64(float xf, const double* __restrict__ t64) {
  const double x = (double)xf;
  const double kd = __builtin_rint(x * 0x1.71547652b82fep+6);
  const int k = (int)kd;
  double r = __builtin_fma(-kd, 0x1.62e42ff000000p-7, x);  // exact: 32-bit constant
  r = __builtin_fma(-kd, -0x1.718432a1b0e26p-41, r);
  double q = 1.0 / 120.0;
  q = __builtin_fma(q, r, 1.0 / 24.0);
  q = __builtin_fma(q, r, 1.0 / 6.0);
  q = __builtin_fma(q, r, 0.5);
  const double p = __builtin_fma(q, r * r, r);
  const double T = t64[k & 63];
  return __builtin_amdgcn_ldexp(__builtin_fma(T, p, T), k >> 6);
}

// (float)exp((double)x), the reference's idiom (ns_core.c:266,278,551,745).
__device__ __forceinline__ float exp_f32_via_f64(float x, const double* __restrict__ t64) {
  const bool in_range = fabsf(x) <= 87.0f;  // false for NaN too
  const double y = exp_lean_f64(in_range ? x : 0.0f, t64);
  const bool ok = in_range && f64_rounds_safely_to_f32(y);
  float r = (float)y;
  if (ASP_NS_RARE(!ok)) r = (float)exp((double)x);
  return r;
}

// (float)tanh((double)a), the reference's idiom (ns_core.c:698,713,725).
//   |a| < 2^-5 : odd series through a^9 (next term < 2^-57 relative)
//   else       : u = exp(-2|a|), tanh = (1 - u) / (1 + u), quotient by a float
//                reciprocal refined twice in fp64 plus one residual correction.
__device__ __forceinline__ float tanh_f32_via_f64(float a, const double* __restrict__ t64) {
  const float ax = fabsf(a);
  const bool finite_ok = ax <= 40.0f;  // beyond: |tanh| rounds to 1, left to libm; NaN too
  const double ad = (double)ax;
  const double a2 = ad * ad;
  double sr = 0x1.664f4882c10fap-6;                 // 62/2835
  sr = __builtin_fma(sr, a2, -0x1.ba1ba1ba1ba1cp-5);  // -17/315
  sr = __builtin_fma(sr, a2, 0x1.1111111111111p-3);   // 2/15
  sr = __builtin_fma(sr, a2, -0x1.5555555555555p-2);  // -1/3
  const double t_small = __builtin_fma(ad * a2, sr, ad);
  const double u = exp_lean_f64(finite_ok ? -2.0f * ax : 0.0f, t64);
  const double num = 1.0 - u, den = 1.0 + u;
  double rc = (double)__builtin_amdgcn_rcpf((float)den);
  rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
  rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
  double qv = num * rc;
  qv = __builtin_fma(__builtin_fma(-qv, den, num), rc, qv);
  const double t = ax < 0.03125f ? t_small : qv;
  const bool ok = finite_ok && f64_rounds_safely_to_f32(t);
  float r = __builtin_copysignf((float)t, a);
  if (ASP_NS_RARE(!ok)) r = (float)tanh((double)a);
  return r;
}

// Correctly rounded sqrtf for x >= 0 (Newton on v_rsq with exact residuals).
// The residual arithmetic underflows below about 2^-102, so inputs under
// 2^-100 (never produced by int16-range audio) take the generic sqrtf; with
// that, equal to sqrtf for every non-negative float (checked exhaustively).
__device__ __forceinline__ float fsqrt(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  float g = x * r;
  float h = 0.5f * r;
  const float e = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, e, g);
  h = __builtin_fmaf(h, e, h);
  const float d = __builtin_fmaf(-g, g, x);
  g = __builtin_fmaf(d, h, g);
  float res = (x == 0.0f || x == __builtin_inff()) ? x : g;
  if (ASP_NS_RARE(x < 0x1p-100f && x > 0.0f)) res = sqrtf(x);
  return res;
}

__device__ __forceinline__ float log_f32_via_f64(float x) {
  const unsigned ax = __float_as_uint(x);
  const bool normal_pos = (ax - 0x00800000u) < 0x7f000000u;  // [2^-126, inf)
  const double y = log_lean_f64(x);
  const bool ok = normal_pos && f64_rounds_safely_to_f32(y);
  float r = (float)y;
  if (__builtin_expect(!ok, 0)) r = (float)log((double)x);
  return r;
}

// Table-driven (float)log((double)x), same contract as log_f32_via_f64 (identical to the ocml form
// for every float, tests/test_ns_gpu.py) at about half the instructions:
//   x = 2^k z with z's bit pattern in [kLogOff, kLogOff + 2^23) (z in [0.697, 1.394));
//   the top 7 pattern bits pick c = 1 / invc (centre of the sub-interval; exactly 1 for the one
//   around 1.0), r = z invc - 1 in one fused step (|r| <= 2^-8), and
//   log x = k ln2 + log c + (r - r^2/2 + ... - r^6/6), truncation < 2^-50 of the result.
// tab[i] = {invc, log c} is built on the host in long double (ns_api.hip).
constexpr unsigned kLogOff = 0x3f328000u;
__device__ __forceinline__ double log_tab_f64(float x, const double2* __restrict__ tab) {
  const unsigned ix = __float_as_uint(x);
  const unsigned tmp = ix - kLogOff;
  const int i = (tmp >> 16) & 127;
  const int k = (int)tmp >> 23;  // arithmetic
  const double z = (double)__uint_as_float(ix - (tmp & 0xff800000u));
  const double2 t = tab[i];
  const double r = __builtin_fma(z, t.x, -1.0);
  const double kd = (double)k;
  double p = -1.0 / 6.0;
  p = __builtin_fma(p, r, 1.0 / 5.0);
  p = __builtin_fma(p, r, -1.0 / 4.0);
  p = __builtin_fma(p, r, 1.0 / 3.0);
  p = __builtin_fma(p, r, -0.5);
  const double hi = __builtin_fma(kd, 0x1.62e42fefa38p-1, t.y);   // 41-bit ln2 head: k ln2_hi exact
  const double lo = __builtin_fma(kd, 0x1.ef35793c7673p-45, __builtin_fma(r * r, p, r));
  return hi + lo;
}
__device__ __forceinline__ float log_f32_via_tab(float x, const double2* __restrict__ tab) {
  const unsigned ax = __float_as_uint(x);
  const bool normal_pos = (ax - 0x00800000u) < 0x7f000000u;  // [2^-126, inf)
  const double y = log_tab_f64(normal_pos ? x : 1.0f, tab);
  const bool ok = normal_pos && f64_rounds_safely_to_f32(y);
  float r = (float)y;
  if (__builtin_expect(!ok, 0)) r = (float)log((double)x);
  return r;
}

// ---- batched forms for the frame kernels -------------------------------------------------------
// N independent arguments per lane: the fast paths run straight-line (no branch between them, so
// the N fp64 chains and their table reads interleave), the per-argument "not decided by the lean
// value" flags are OR-ed and ONE rarely taken branch redoes the lane's arguments through libm (out
// of line).  Results equal the one-argument forms above.
// libm forms out of line: the rarely taken branches call them instead of carrying N inlined copies
__device__ __attribute__((noinline)) float log_f32_slow(float x) { return (float)log((double)x); }
__device__ __attribute__((noinline)) float exp_f32_slow(float x) { return (float)exp((double)x); }
__device__ __attribute__((noinline)) float sqrt_f32_slow(float x) { return sqrtf(x); }

template <int N>
__device__ __forceinline__ void log_f32_via_tab_n(const float (&x)[N], float (&out)[N],
                                                  const double2* __restrict__ tab) {
  unsigned bad = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    // an argument outside [2^-126, inf) runs through the lean form as it is (the table index is masked,
    // nothing can fault) and is flagged: the fallback below recomputes the lane's values
    const unsigned ax = __float_as_uint(x[k]);
    const unsigned normal_pos = (ax - 0x00800000u) < 0x7f000000u ? 1u : 0u;  // [2^-126, inf)
    const double y = log_tab_f64(x[k], tab);
    bad |= (normal_pos ^ 1u) | (f64_rounds_safely_to_f32(y) ? 0u : 1u);
    out[k] = (float)y;
  }
  if (ASP_NS_RARE(bad != 0)) {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = log_f32_slow(x[k]);
  }
}

template <int N>
__device__ __forceinline__ void exp_f32_via_f64_n(const float (&x)[N], float (&out)[N],
                                                  const double* __restrict__ t64) {
  unsigned bad = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    // out-of-range arguments run through the lean form as they are (masked table index) and are flagged
    const unsigned in_range = fabsf(x[k]) <= 87.0f ? 1u : 0u;  // false for NaN too
    const double y = exp_lean_f64(x[k], t64);
    bad |= (in_range ^ 1u) | (f64_rounds_safely_to_f32(y) ? 0u : 1u);
    out[k] = (float)y;
  }
  if (ASP_NS_RARE(bad != 0)) {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = exp_f32_slow(x[k]);
  }
}

// fsqrt for N arguments, one merged check for the tiny-argument case
template <int N>
__device__ __forceinline__ void fsqrt_n(const float (&x)[N], float (&out)[N]) {
  unsigned bad = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float xv = x[k];
    const float r = __builtin_amdgcn_rsqf(xv);
    float g = xv * r;
    float h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-g, g, xv);
    g = __builtin_fmaf(d, h, g);
    out[k] = (xv == 0.0f || xv == __builtin_inff()) ? xv : g;
    bad |= (xv < 0x1p-100f && xv > 0.0f) ? 1u : 0u;
  }
  if (ASP_NS_RARE(bad != 0)) {
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (x[k] < 0x1p-100f && x[k] > 0.0f) out[k] = sqrt_f32_slow(x[k]);
  }
}

// ---- lean fp64 forms of the echo canceller's float transcendentals (aec_core.c:280, 487-488) ----
// Each returns exactly what its plain form `(float)f((double)x)` (ocml fp64, rounded) returns: the
// lean evaluation is accurate to ~2^-46, the float rounding is taken from it only when the fp64
// value sits more than 2^-44 (relative) away from a rounding boundary, everything else -- and every
// argument outside the fast range -- goes to the plain form.  Compared with the plain forms over
// whole float ranges on the device in tests/test_aec_gpu.py.

// exp of a double argument with |x| <= 87 (float result normal): exp_lean_f64 for a double.
__device__ __forceinline__ double exp_lean_f64d(double x, const double* __restrict__ t64) {
  const double kd = __builtin_rint(x * 0x1.71547652b82fep+6);
  const int k = (int)kd;
  double r = __builtin_fma(-kd, 0x1.62e42ff000000p-7, x);
  r = __builtin_fma(-kd, -0x1.718432a1b0e26p-41, r);
  double q = 1.0 / 120.0;
  q = __builtin_fma(q, r, 1.0 / 24.0);
  q = __builtin_fma(q, r, 1.0 / 6.0);
  q = __builtin_fma(q, r, 0.5);
  const double p = __builtin_fma(q, r * r, r);
  const double T = t64[k & 63];
  return __builtin_amdgcn_ldexp(__builtin_fma(T, p, T), k >> 6);
}

// (float)pow((double)x, (double)y).  Fast range: x a positive normal float, |y log x| <= 40.
__device__ __forceinline__ float pow_f32_via_f64(float x, float y, const double* __restrict__ t64) {
  const unsigned ax = __float_as_uint(x);
  const bool normal_pos = (ax - 0x00800000u) < 0x7f000000u;
  const double t = (double)y * log_lean_f64(normal_pos ? x : 1.0f);
  const bool in_range = normal_pos && (__builtin_fabs(t) <= 40.0);  // false for NaN too
  const double v = exp_lean_f64d(in_range ? t : 0.0, t64);
  const bool ok = in_range && f64_rounds_safely_to_f32(v);
  float r = (float)v;
  if (__builtin_expect(!ok, 0)) r = (float)pow((double)x, (double)y);
  return r;
}

// (float)cos((double)x) and (float)sin((double)x) for one argument.  Fast range: 0 <= x < 8:
// k = rint(x 2/pi), r = x - k pi/2 in two exact-enough steps (33-bit head: x - k C1 is exact for a
// float x, then one fused step with the next 53 bits), kernels of fdlibm (k_sin.c / k_cos.c).
__device__ __forceinline__ void sincos_f32_via_f64(float x, float& s_out, float& c_out) {
  const bool in_range = __float_as_uint(x) < 0x41000000u;  // [+0, 8): not -0, not NaN
  const double xd = in_range ? (double)x : 0.0;
  const double kd = __builtin_rint(xd * 0x1.45f306dc9c883p-1);  // 2/pi
  const int k = (int)kd;
  double r = __builtin_fma(-kd, 0x1.921fb54400000p+0, xd);       // pi/2, 33 significant bits: exact
  r = __builtin_fma(-kd, 0x1.0b4611a626331p-34, r);              // next 53 bits
  const double z = r * r;
  // k_sin
  double ps = 1.58969099521155010221e-10;                          // S6
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);          // S5
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);           // S4
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);          // S3
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);           // S2
  const double v3 = z * r;
  const double sn = __builtin_fma(v3, __builtin_fma(z, ps, -1.66666666666666324348e-01), r);  // S1
  // k_cos
  double pc = -1.13596475577881948265e-11;                         // C6
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);           // C5
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);          // C4
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);           // C3
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);          // C2
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);           // C1
  const double hz = 0.5 * z;
  const double a = 1.0 - hz;
  const double cs = a + (((1.0 - a) - hz) + z * z * pc);
  const int q = k & 3;
  const double sv = (q & 1) ? cs : sn, cv = (q & 1) ? sn : cs;
  const double sd = (q & 2) ? -sv : sv;
  const double cd = ((q + 1) & 2) ? -cv : cv;
  const bool ok = in_range && f64_rounds_safely_to_f32(sd) && f64_rounds_safely_to_f32(cd);
  s_out = (float)sd;
  c_out = (float)cd;
  if (__builtin_expect(!ok, 0)) {
    s_out = (float)sin((double)x);
    c_out = (float)cos((double)x);
  }
}

// --------------------------------------------------------------------------
// Histogram window close: FeatureParameterExtraction(self, 1), ns_core.c:337-517.
// Runs once per 500 frames per stream.  Zero bins cannot change any of the
// running sums / peaks, so only non-empty bins are visited, in bin order, which
// keeps the reference's sequential float sums and tie-breaking exactly.
struct PriorModel {
  float p0, p1, p3, p4, p5, p6;
};

// FLOW (ns_kernels1.hip's hand-off build): the histogram is read and cleared with agent-scope (sc1)
// accesses, like every other state access of that build.
template <bool FLOW>
__device__ __forceinline__ int hist_ld(const int32_t* p) {
  typedef __attribute__((address_space(1))) int gi32;
  if constexpr (FLOW) return __hip_atomic_load((const gi32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool FLOW>
__device__ __forceinline__ void hist_st(int32_t* p, int v) {
  typedef __attribute__((address_space(1))) int gi32;
  if constexpr (FLOW) __hip_atomic_store((gi32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

template <bool FLOW = false>
__device__ inline __attribute__((noinline)) PriorModel close_histogram_window(int32_t* __restrict__ hist, int lane,
                                                          int updateWindow, bool zero_after,
                                                          PriorModel pm) {
  // ---- LRT histogram, :340-373
  float avgHistLrt = 0.f, avgHistLrtCompl = 0.f, avgSquareHistLrt = 0.f;
  int numHistLrt = 0;
  for (int r = 0; r < 16; ++r) {
    const int i = r * 64 + lane;
    const int v = i < kHist ? hist_ld<FLOW>(hist + i) : 0;
    unsigned long long m = __ballot(v != 0);
    while (m) {
      const int p = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int hv = __shfl(v, p, 64);
      const float binMid = ((float)(r * 64 + p) + 0.5f) * 0.1f;
      if (binMid <= 1.f) {
        avgHistLrt += hv * binMid;
        numHistLrt += hv;
      }
      avgSquareHistLrt += hv * binMid * binMid;
      avgHistLrtCompl += hv * binMid;
    }
  }
  if (numHistLrt > 0) avgHistLrt = avgHistLrt / ((float)numHistLrt);
  avgHistLrtCompl = avgHistLrtCompl / ((float)updateWindow);
  avgSquareHistLrt = avgSquareHistLrt / ((float)updateWindow);
  const float fluctLrt = avgSquareHistLrt - avgHistLrt * avgHistLrtCompl;
  if (fluctLrt < 0.05f) {
    pm.p0 = 1.f;
  } else {
    pm.p0 = 1.2f * avgHistLrt;
    if (pm.p0 < 0.2f) pm.p0 = 0.2f;
    if (pm.p0 > 1.f) pm.p0 = 1.f;
  }
  // ---- two dominant peaks of the flatness and difference histograms, :378-432
  float pos1[2], pos2[2];
  int wt1[2], wt2[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int32_t* hh = hist + (k + 1) * kHistStride;
    const float binSize = k == 0 ? 0.05f : 0.1f;
    int maxPeak1 = 0, maxPeak2 = 0;
    pos1[k] = 0.f;
    pos2[k] = 0.f;
    wt1[k] = 0;
    wt2[k] = 0;
    for (int r = 0; r < 16; ++r) {
      const int i = r * 64 + lane;
      const int v = i < kHist ? hist_ld<FLOW>(hh + i) : 0;
      unsigned long long m = __ballot(v != 0);
      while (m) {
        const int p = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int hv = __shfl(v, p, 64);
        const float binMid = ((float)(r * 64 + p) + 0.5f) * binSize;
        if (hv > maxPeak1) {
          maxPeak2 = maxPeak1;
          wt2[k] = wt1[k];
          pos2[k] = pos1[k];
          maxPeak1 = hv;
          wt1[k] = hv;
          pos1[k] = binMid;
        } else if (hv > maxPeak2) {
          maxPeak2 = hv;
          wt2[k] = hv;
          pos2[k] = binMid;
        }
      }
    }
  }
  const int thresWeight = (int)(0.3 * updateWindow);  // :67-70
  // ---- flatness, :435-463
  int useFlat = 1;
  if ((fabsf(pos2[0] - pos1[0]) < 2 * 0.05f) && (wt2[0] > 0.5f * wt1[0])) {
    wt1[0] += wt2[0];
    pos1[0] = 0.5f * (pos1[0] + pos2[0]);
  }
  if (wt1[0] < thresWeight || pos1[0] < 0.6f) useFlat = 0;
  if (useFlat == 1) {
    pm.p1 = 0.9f * pos1[0];
    if (pm.p1 < 0.1f) pm.p1 = 0.1f;
    if (pm.p1 > 0.95f) pm.p1 = 0.95f;
  }
  // ---- template difference, :467-498
  int useDiff = 1;
  if ((fabsf(pos2[1] - pos1[1]) < 2 * 0.1f) && (wt2[1] > 0.5f * wt1[1])) {
    wt1[1] += wt2[1];
    pos1[1] = 0.5f * (pos1[1] + pos2[1]);
  }
  pm.p3 = 1.2f * pos1[1];
  if (wt1[1] < thresWeight) useDiff = 0;
  if (pm.p3 < 0.16f) pm.p3 = 0.16f;
  if (pm.p3 > 1.f) pm.p3 = 1.f;
  if (fluctLrt < 0.05f) useDiff = 0;
  const float featureSum = (float)(1 + useFlat + useDiff);  // :504-507
  pm.p4 = 1.f / featureSum;
  pm.p5 = ((float)useFlat) / featureSum;
  pm.p6 = ((float)useDiff) / featureSum;
  if (zero_after) {  // :510-516
    for (int k = 0; k < 3; ++k)
      for (int r = 0; r < 16; ++r) hist_st<FLOW>(hist + k * kHistStride + r * 64 + lane, 0);
  }
  return pm;
}


}  // namespace aspns_dev
