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

// Half of one radix-4 butterfly of cft1st / cftmdl (fft4g.c:1002-1231): lane parity h = 0 produces
// outputs 0 and 2, h = 1 outputs 1 and 3.  With s = (h ? -1 : +1):
//   u = c0 + s c1, v = c2 + s c3, h: v <- i v, p = u + v, m = u - v, first = tA p, second = tB m
// (tA, tB from the per-lane table: identity for twiddle-free blocks; `diag` selects the
// reference's factored form of the w[2] block).  Same operations as ns_kernels.hip's cft_half_pass.
__device__ __forceinline__ void half_bfly(float2 c0, float2 c1, float2 c2, float2 c3, bool h,
                                          float4 tw, bool diag, float2& first, float2& second) {
  const uint32_t sm = h ? 0x80000000u : 0u;
  const float ur = c0.x + xorf(c1.x, sm), ui = c0.y + xorf(c1.y, sm);
  const float vr = c2.x + xorf(c3.x, sm), vi = c2.y + xorf(c3.y, sm);
  const float vr2 = h ? -vi : vr;
  const float vi2 = h ? vr : vi;
  const float pr = ur + vr2, pi = ui + vi2;
  const float mr = ur - vr2, mi = ui - vi2;
  const float g1r = tw.x * pr - tw.y * pi, g1i = tw.x * pi + tw.y * pr;
  const float g2r = tw.z * mr - tw.w * mi, g2i = tw.z * mi + tw.w * mr;
  const float d1r = tw.x * (pr - pi), d1i = tw.x * (pr + pi);
  const float d2r = -(tw.x * (mr + mi)), d2i = tw.x * (mr - mi);
  first = diag ? make_float2(d1r, d1i) : make_float2(g1r, g1i);
  second = diag ? make_float2(d2r, d2i) : make_float2(g2r, g2i);
}

// Passes 1-3 of cftfsub / cftbsub for 128 complex points.  In: tile holds the inputs in natural
// order (pass 1 reads bit-reversed = bitrv2).  Out: oA = element q + 64 g + 16 h, oB = oA's + 32.
__device__ __forceinline__ void cft128_passes1(float2* tile, const float* tws, int diagbits,
                                               int lane, float2& oA, float2& oB) {
  const int b = lane >> 1;
  const bool h = (lane & 1) != 0;
  float2 f, s;
  {
    const int rb = (int)(__brev((unsigned)b) >> 27);
    const float4 tw = *reinterpret_cast<const float4*>(tws + (0 * 64 + lane) * 4);
    half_bfly(tile[rb], tile[rb + 64], tile[rb + 32], tile[rb + 96], h, tw, (diagbits & 1) != 0, f, s);
    lds_sync1();
    tile[4 * b + (h ? 1 : 0)] = f;
    tile[4 * b + (h ? 3 : 2)] = s;
  }
  lds_sync1();
  {
    const int base = 16 * (b >> 2) + (b & 3);
    const float4 tw = *reinterpret_cast<const float4*>(tws + (1 * 64 + lane) * 4);
    half_bfly(tile[base], tile[base + 4], tile[base + 8], tile[base + 12], h, tw, (diagbits & 2) != 0, f, s);
    lds_sync1();
    tile[base + (h ? 4 : 0)] = f;
    tile[base + (h ? 12 : 8)] = s;
  }
  lds_sync1();
  {
    const int base = 64 * (b >> 4) + (b & 15);
    const float4 tw = *reinterpret_cast<const float4*>(tws + (2 * 64 + lane) * 4);
    half_bfly(tile[base], tile[base + 16], tile[base + 32], tile[base + 48], h, tw, (diagbits & 4) != 0, oA, oB);
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
__device__ __forceinline__ void radix2_tail1(float2& a, float2& b, uint32_t gmask, bool backward) {
  const float ar = tail_combine(a.x, gmask), ai = tail_combine(a.y, gmask);
  const float br = tail_combine(b.x, gmask), bi = tail_combine(b.y, gmask);
  a = make_float2(ar, backward ? -ai : ai);
  b = make_float2(br, backward ? -bi : bi);
}

// rftfsub / rftbsub (fft4g.c:1234-1283) plus the a[0]/a[1] fix-ups of rdft (fft4g.c:347-352):
// element E = q + 16 t + 64 g pairs with 128 - E; g = 0 lanes hold the j side, g = 1 the k side.
__device__ __forceinline__ void real_split1(float2* tile, const float* spls, int lane, float2 e[2],
                                            bool backward) {
  const int b = lane >> 1, h = lane & 1;
  const bool hi = b >= 16;
  const int base = 64 * (b >> 4) + (b & 15) + 16 * h;
  lds_sync1();
  tile[base] = e[0];
  tile[base + 32] = e[1];
  lds_sync1();
  // J - K and J + K of the reference (J = the j side, K = the k side) without selecting which is which:
  // xr = J.x - K.x = +-(e.x - pe.x) with the sign of the side (a - b == -(b - a) exactly), xi = J.y + K.y
  // = e.y + pe.y on either side; the updates e.x -+ yr likewise take yr's sign from the side.
  const uint32_t himask = hi ? 0x80000000u : 0u;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int E = base + 32 * k;
    const float2 pe = tile[(128 - E) & 127];
    const float2 w = *reinterpret_cast<const float2*>(spls + (b * 4 + h + 2 * k) * 2);  // (wkr, wki)
    const float xr = xorf(e[k].x - pe.x, himask), xi = e[k].y + pe.y;
    float2 r;
    if (!backward) {
      const float yr = w.x * xr - w.y * xi, yi = w.x * xi + w.y * xr;
      r = make_float2(e[k].x - xorf(yr, himask), e[k].y - yi);
      if (k == 0) {  // elements 0 and 64 are slot 0 of lanes 0 and 32
        if (E == 0) r = make_float2(e[k].x + e[k].y, e[k].x - e[k].y);
        if (E == 64) r = e[k];
      }
    } else {
      const float yr = w.x * xr + w.y * xi, yi = w.x * xi - w.y * xr;
      r = make_float2(e[k].x - xorf(yr, himask), yi - e[k].y);
      if (k == 0) {
        if (E == 0) {
          const float hh = 0.5f * (e[k].x - e[k].y);
          r = make_float2(e[k].x - hh, -hh);
        }
        if (E == 64) r = make_float2(e[k].x, -e[k].y);
      }
    }
    e[k] = r;
  }
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


}  // namespace aspns_pair
