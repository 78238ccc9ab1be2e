// ns_pair_fft.h -- device helpers of the one-stream-per-wave pair-layout frame kernel
// (ns_kernels1.hip): the 128-point complex transform with half a radix-4 butterfly
// per lane, its radix-2 tail and real split, the PCM store and the scalar-row lane write.
// Lane L = 2 lam + h (lam = q + 16 g) owns elements E = q + 64 g + 16 h and E + 32.
#pragma once
#include <hip/hip_runtime.h>

#include "ns_device.h"
#include "ns_layout.h"

namespace aspns_pair {
using namespace aspns_dev;

__device__ __forceinline__ void lds_sync1() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane constants of the transform, derived once per kernel from the lane index and the table's
// diag bits: s = +1 / -1 by lane parity, kap[p] = 1 on the lanes whose pass-p butterfly is the
// reference's factored w[2] block (fft4g.c:1044-1058, 1163-1177), 0 elsewhere.
struct PairFftLane {  // scalar members only: an array member is kept in memory (LDS) by the compiler
  float s;
  float kap0, kap1, kap2;
  bool h;
};
__device__ __forceinline__ PairFftLane pair_fft_lane(int lane, int diagbits) {
  PairFftLane L;
  L.h = (lane & 1) != 0;
  L.s = L.h ? -1.0f : 1.0f;
  L.kap0 = (float)(diagbits & 1);
  L.kap1 = (float)((diagbits >> 1) & 1);
  L.kap2 = (float)((diagbits >> 2) & 1);
  return L;
}

// {k a.y' + a.x, k a.x' + a.y} with (a.y', a.x') = (-a.y, a.x) (ROT = +1: a + k i a) or (a.y, -a.x)
// (ROT = -1: a - k i a); k = 0 returns a (x + (+-0) == x), k = 1 the sum / difference pair of the
// reference's factored butterfly.  One v_pk_fma_f32 (1 x and 0 x are exact, so each half is one rounding
// of the reference's own addition).
template <int ROT>
__device__ __forceinline__ f32x2 rot_add(float k, f32x2 a) {
  f32x2 kk = {k, k}, r;
  if (ROT > 0)
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(kk), "v"(a));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(kk), "v"(a));
  return r;
}
// {a.x t.x - a.y t.y, a.x t.y + a.y t.x}: the complex product in the reference's operation order
__device__ __forceinline__ f32x2 cmul_pk(f32x2 a, f32x2 t) {
  const f32x2 p1 = a.xx * t;     // {a.x t.x, a.x t.y}
  const f32x2 p2 = a.yy * t.yx;  // {a.y t.y, a.y t.x}
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p1), "v"(p2));
  return r;
}

// Half of one radix-4 butterfly of cft1st / cftmdl (fft4g.c:1002-1231): lane parity h = 0 produces
// outputs 0 and 2, h = 1 outputs 1 and 3.  With s = (h ? -1 : +1):
//   u = c0 + s c1, v = c2 + s c3, h: v <- i v, p = u + v, m = u - v, first = tA p', second = tB m'
// (tA, tB from the per-lane table: identity for twiddle-free blocks).  The reference's factored form of
// the w[2] block -- w (p.r - p.i), w (p.r + p.i) and -w (m.r + m.i), w (m.r - m.i) -- is the same
// product with p' = p + i p, m' = m - i m, tA = (w, 0), tB = (-w, 0): rot_add under the lane's kap,
// no branch and no select (a 0 x term adds +-0).  Same operations as ns_kernels.hip's cft_half_pass.
__device__ __forceinline__ void half_bfly(f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3, const PairFftLane& L,
                                          float kap, float4 tw, f32x2& first, f32x2& second) {
  const f32x2 ss = {L.s, L.s};
  const f32x2 u = __builtin_elementwise_fma(ss, c1, c0);
  const f32x2 v = __builtin_elementwise_fma(ss, c3, c2);
  const f32x2 v2 = {L.h ? -v.y : v.x, L.h ? v.x : v.y};
  const f32x2 p = u + v2, m = u - v2;
  first = cmul_pk(rot_add<+1>(kap, p), f32x2{tw.x, tw.y});
  second = cmul_pk(rot_add<-1>(kap, m), f32x2{tw.z, tw.w});
}

__device__ __forceinline__ f32x2 ldsc(const float2* t, int i) {
  const float2 v = t[i];
  return f32x2{v.x, v.y};
}
__device__ __forceinline__ void stsc(float2* t, int i, f32x2 v) { t[i] = make_float2(v.x, v.y); }

// Passes 1-3 of cftfsub / cftbsub for 128 complex points.  In: tile holds the inputs in natural
// order (pass 1 reads bit-reversed = bitrv2).  Out: oA = element q + 64 g + 16 h, oB = oA's + 32.
__device__ __forceinline__ void cft128_passes1(float2* tile, const float* tws, const PairFftLane& L,
                                               int lane, f32x2& oA, f32x2& oB) {
  const int b = lane >> 1;
  const int hh = lane & 1;
  f32x2 f, s;
  {
    const int rb = (int)(__brev((unsigned)b) >> 27);
    const float4 tw = *reinterpret_cast<const float4*>(tws + (0 * 64 + lane) * 4);
    half_bfly(ldsc(tile, rb), ldsc(tile, rb + 64), ldsc(tile, rb + 32), ldsc(tile, rb + 96), L, L.kap0, tw, f, s);
    lds_sync1();
    stsc(tile, 4 * b + hh, f);
    stsc(tile, 4 * b + 2 + hh, s);
  }
  lds_sync1();
  {
    const int base = 16 * (b >> 2) + (b & 3);
    const float4 tw = *reinterpret_cast<const float4*>(tws + (1 * 64 + lane) * 4);
    half_bfly(ldsc(tile, base), ldsc(tile, base + 4), ldsc(tile, base + 8), ldsc(tile, base + 12), L, L.kap1, tw, f, s);
    lds_sync1();
    stsc(tile, base + 4 * hh, f);
    stsc(tile, base + 8 + 4 * hh, s);
  }
  lds_sync1();
  {
    const int base = 64 * (b >> 4) + (b & 15);
    const float4 tw = *reinterpret_cast<const float4*>(tws + (2 * 64 + lane) * 4);
    half_bfly(ldsc(tile, base), ldsc(tile, base + 16), ldsc(tile, base + 32), ldsc(tile, base + 48), L, L.kap2, tw, oA, oB);
  }
}

// radix-2 tail (fft4g.c:939-947 / 989-997): element p (lanes < 32) with p + 64 (lanes >= 32, same
// lane & 31).  v_permlane32_swap on two copies leaves the low half's value on both halves of one
// copy and the high half's on the other; p: a + c, p + 64: a - c, i.e. a + (+-c) on both.
__device__ __forceinline__ float tail_combine(float v, uint32_t gmask) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + xorf(__uint_as_float(r[1]), gmask);
}
// out: the two elements' real parts / imaginary parts as pairs {slot 0, slot 1}
__device__ __forceinline__ void radix2_tail1(f32x2 a, f32x2 b, uint32_t gmask, bool backward, f32x2& er, f32x2& ei) {
  const float ar = tail_combine(a.x, gmask), ai = tail_combine(a.y, gmask);
  const float br = tail_combine(b.x, gmask), bi = tail_combine(b.y, gmask);
  er = f32x2{ar, br};
  ei = backward ? f32x2{-ai, -bi} : f32x2{ai, bi};
}

// rftfsub / rftbsub (fft4g.c:1234-1283) plus the a[0]/a[1] fix-ups of rdft (fft4g.c:347-352):
// element E = q + 16 t + 64 g pairs with 128 - E; g = 0 lanes hold the j side, g = 1 the k side.
// In / out: the lane's two elements as pairs {slot 0, slot 1} of real and of imaginary parts.  Elements
// 0 and 64 pair with themselves; their table factors are 0, so the general form returns element 64
// unchanged (conjugated on the way back), and element 0 (slot 0 of lane 0) takes the rdft fix-up.  Lane 0
// reads its slot-0 partner from position 128 - 0: the first slot of the next tile row, any finite or
// non-finite garbage -- the fix-up replaces what was computed from it.
__device__ __forceinline__ void real_split1(float2* tile, const float* spls, int lane, f32x2& er, f32x2& ei,
                                            bool backward) {
  const int b = lane >> 1, h = lane & 1;
  const bool hi = b >= 16;
  const int base = 64 * (b >> 4) + (b & 15) + 16 * h;
  float* tf = reinterpret_cast<float*>(tile);
  lds_sync1();
  tf[2 * base] = er.x;
  tf[2 * base + 1] = ei.x;
  tf[2 * base + 64] = er.y;
  tf[2 * base + 65] = ei.y;
  lds_sync1();
  // J - K and J + K of the reference (J = the j side, K = the k side) without selecting which is which:
  // xr = J.x - K.x = +-(e.x - pe.x) with the sign of the side (a - b == -(b - a) exactly), xi = J.y + K.y
  // = e.y + pe.y on either side; the updates e.x -+ yr likewise take yr's sign from the side.
  const float sg = hi ? -1.0f : 1.0f;
  const f32x2 sgs = {sg, sg};
  const int pi0 = 2 * (128 - base);  // partner of slot 0; slot 1's is 32 elements below
  const f32x2 per = {tf[pi0], tf[pi0 - 64]}, pei = {tf[pi0 + 1], tf[pi0 - 63]};
  const float* wp = spls + (b * 4 + h) * 2;  // (wkr, wki) of slot 0; slot 1's entry is two further on
  const f32x2 wr = {wp[0], wp[4]}, wi = {wp[1], wp[5]};
  const f32x2 xr = sgs * (er - per), xi = ei + pei;
  f32x2 rr, ri;
  if (!backward) {
    const f32x2 yr = wr * xr - wi * xi, yi = wr * xi + wi * xr;
    rr = __builtin_elementwise_fma(-sgs, yr, er);  // e.x -+ yr: (-+1) yr is exact, one rounding
    ri = ei - yi;
    if (lane == 0) {
      rr.x = er.x + ei.x;
      ri.x = er.x - ei.x;
    }
  } else {
    const f32x2 yr = wr * xr + wi * xi, yi = wr * xi - wi * xr;
    rr = __builtin_elementwise_fma(-sgs, yr, er);
    ri = yi - ei;
    if (lane == 0) {
      const float hh = 0.5f * (er.x - ei.x);
      rr.x = er.x - hh;
      ri.x = -hh;
    }
  }
  er = rr;
  ei = ri;
}

template <bool IO16>
__device__ __forceinline__ void store2p(float* y, int idx, float a, float b) {
  if (IO16) {
    short2 v;
    const float kMaxRound = 32767 - 0.5f, kMinRound = -32768 + 0.5f;
    v.x = a > 0 ? (a >= kMaxRound ? (short)32767 : (short)(a + 0.5f))
                : (a <= kMinRound ? (short)-32768 : (short)(a - 0.5f));
    v.y = b > 0 ? (b >= kMaxRound ? (short)32767 : (short)(b + 0.5f))
                : (b <= kMinRound ? (short)-32768 : (short)(b - 0.5f));
    *reinterpret_cast<short2*>(reinterpret_cast<short*>(y) + idx) = v;
  } else {
    *reinterpret_cast<float2*>(y + idx) = make_float2(a, b);
  }
}

__device__ __forceinline__ float sat16p(float x) {
  return x > 32767 ? 32767 : (x < -32768 ? -32768 : x);
}

// v_writelane_b32: the (wave-uniform) value goes into lane K of the scalar row
template <int K>
__device__ __forceinline__ float writelane_bits(float row, int bits) {
  int r = __float_as_int(row);
  const int u = __builtin_amdgcn_readfirstlane(bits);
  asm("v_writelane_b32 %0, %1, %2" : "+v"(r) : "s"(u), "n"(K));
  return __int_as_float(r);
}
// The same for a wave-uniform value that lives in a VGPR (the result of vector float arithmetic): one
// v_cndmask_b32 under a constant lane mask instead of v_readfirstlane + v_writelane (every lane holds
// the value, lane K keeps it).
template <int K>
__device__ __forceinline__ float setlane_vgpr(float row, float val) {
  const unsigned long long mask = 1ull << K;
  asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(row) : "v"(val), "s"(mask));
  return row;
}


}  // namespace aspns_pair
