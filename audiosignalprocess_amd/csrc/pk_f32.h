// pk_f32.h -- packed-f32 helpers shared by the BlockThresholding and echo-canceller kernels: a complex
// value is one 64-bit register pair (x = re, y = im), so complex adds and multiplies issue as
// v_pk_add_f32 / v_pk_mul_f32 (one issue slot for both halves).  The mixed-sign and swapped forms use
// the VOP3P neg_lo / neg_hi / op_sel modifiers, which the compiler does not form from per-lane
// negations, through one-instruction asm statements.  Every half is the IEEE single-precision
// operation of its scalar spelling (x - y == x + (-y), (-x) + y == y - x bit for bit), so results
// equal those of the reference's scalar code.
#pragma once
#include <hip/hip_runtime.h>

namespace asppk {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ASP_PK_ADD(name, mods)                                              \
  __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b) {                 \
    f32x2 r;                                                                \
    asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));       \
    return r;                                                               \
  }
ASP_PK_ADD(add_sub_lo, "neg_lo:[0,1]")                                       // {a.x - b.x, a.y + b.y}
ASP_PK_ADD(add_sub_hi, "neg_hi:[0,1]")                                       // {a.x + b.x, a.y - b.y}
ASP_PK_ADD(add_neg0_hi, "neg_hi:[1,0]")                                      // {a.x + b.x, -a.y + b.y}
ASP_PK_ADD(add_neg_both_hi, "neg_hi:[1,1]")                                  // {a.x + b.x, -a.y - b.y}
ASP_PK_ADD(sub_lo_rsub_hi, "neg_lo:[0,1] neg_hi:[1,0]")                      // {a.x - b.x, b.y - a.y}
ASP_PK_ADD(add_swap, "op_sel:[0,1] op_sel_hi:[1,0]")                         // {a.x + b.y, a.y + b.x}
ASP_PK_ADD(sub_swap, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")  // {a.x - b.y, a.y - b.x}
ASP_PK_ADD(add_swap_sub_hi, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")     // {a.x + b.y, a.y - b.x}
ASP_PK_ADD(add_swap_sub_lo, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")     // {a.x - b.y, a.y + b.x}
ASP_PK_ADD(swap0_add_sub_hi, "op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[0,1]")    // {a.y + b.x, a.x - b.y}
#undef ASP_PK_ADD
// {a.x - a.y, a.x + a.y} and {a.y - a.x, a.y + a.x}
__device__ __forceinline__ f32x2 diff_sum(f32x2 a) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ f32x2 rdiff_sum(f32x2 a) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a));
  return r;
}

// {a.r t.r - a.i t.i, a.r t.i + a.i t.r} (C_MUL(a, t)); CONJ: a times the conjugate of t, as a
// multiplication by {t.r, -t.i} computes it: {a.r t.r - a.i (-t.i), a.r (-t.i) + a.i t.r}
template <bool CONJ>
__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 t) {
  const f32x2 p1 = a.xx * t;     // {a.r t.r, a.r t.i}
  const f32x2 p2 = a.yy * t.yx;  // {a.i t.i, a.i t.r}
  return CONJ ? add_neg0_hi(p1, p2) : add_sub_lo(p1, p2);
}
// {w.x v.x + w.y v.y, w.x v.y - w.y v.x}
__device__ __forceinline__ f32x2 cmul_conj_w(f32x2 w, f32x2 v) {
  const f32x2 p1 = w.xx * v;
  const f32x2 p2 = w.yy * v.yx;
  return add_sub_hi(p1, p2);
}

}  // namespace asppk
