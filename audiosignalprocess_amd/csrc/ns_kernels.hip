// ns_kernels.hip -- hand-written gfx950 kernels of the batched noise suppressor.
//
// Replaces, for N independent streams per launch (16 kHz geometry: 160 / 256 / 129; template flag G8:
// the 8 kHz geometry 80 / 128 / 65 of ns_core.c:89-98), the reference's
//   WebRtcNs_AnalyzeCore  (ns/ns_core.c:1043-1181)
//   WebRtcNs_ProcessCore  (ns/ns_core.c:1183-1359)
//   WebRtc_rdft(256, +-1) (utility/fft4g.c:324-362)
// (paths relative to WebRtc_AMP_Port/webrtc/modules/audio_processing/).
//
// Mapping: one wave64 per stream, four streams per 256-thread workgroup, no
// workgroup barrier anywhere.  Lane q owns bins q and q+64; bin 128 is computed
// redundantly on every lane.  Per-stream scalars are wave-uniform, so every
// data-independent branch of the reference (startup phases, tracker publish,
// histogram window) is a scalar branch.
//
// FFT: the 256-point real transform runs as a 128-point complex transform whose
// three radix-4 passes are staged through a 1 KB LDS tile private to the wave;
// each radix-4 butterfly is split over a lane pair (one lane produces outputs
// 0/2, the other 1/3) so all 64 lanes work; the radix-2 tail and the real
// split use registers plus one wavefront shuffle (lane q <-> lane 64-q).  The
// butterflies perform the float operations of the reference's Ooura code in
// the same order, so spectra are bit-identical to WebRtc_rdft.
//
// Cross-bin sums use a fixed association (slot-local, then xor 1,2,4,8,16,32)
// that oracle/ns_oracle.c reproduces in ASP_NS_REDUCE_TREE mode.
//
// Compile with -ffp-contract=off: parity depends on unfused mul/add.
#include <hip/hip_runtime.h>

#include "ns_device.h"
#include "ns_layout.h"

using namespace aspns;

namespace {

using namespace aspns_dev;


struct FftLane {
  float4 tw0, tw1, tw2;  // (tAr, tAi, tBr, tBi) for passes 1..3
  int diag;
  float cq, cr;
};

__device__ __forceinline__ FftLane load_fft_lane(const NsTables* __restrict__ T, int lane) {
  FftLane L;
  L.tw0 = *reinterpret_cast<const float4*>(T->tw[0][lane]);
  L.tw1 = *reinterpret_cast<const float4*>(T->tw[1][lane]);
  L.tw2 = *reinterpret_cast<const float4*>(T->tw[2][lane]);
  L.diag = T->diag[lane];
  L.cq = T->cq[lane];
  L.cr = T->cr[lane];
  return L;
}

// Half of one radix-4 butterfly of cft1st / cftmdl (fft4g.c:1002-1231).
// Lane parity h = 0 produces outputs 0 and 2 of the butterfly, h = 1 outputs
// 1 and 3.  With s = (h ? -1 : +1):
//   u = c0 + s*c1, v = c2 + s*c3, h: v <- i*v, p = u + v, m = u - v
//   out_first = tA * p, out_second = tB * m
// where tA/tB come from the per-lane table (identity for twiddle-free blocks)
// and `diag` selects the reference's factored form for the w[2] block.
__device__ __forceinline__ void cft_half_pass(float2* buf, int r0, int r1, int r2, int r3,
                                              int w0, int w1, bool h, float4 tw, bool diag) {
  const float2 c0 = buf[r0], c1 = buf[r1], c2 = buf[r2], c3 = buf[r3];
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
  buf[w0] = diag ? make_float2(d1r, d1i) : make_float2(g1r, g1i);
  buf[w1] = diag ? make_float2(d2r, d2i) : make_float2(g2r, g2i);
}

// The three radix-4 passes of cftfsub/cftbsub for 128 complex points; input in
// natural order in buf (the first pass reads bit-reversed = bitrv2).
__device__ __forceinline__ void cft128_passes(float2* buf, const FftLane& L, int lane) {
  const int b = lane >> 1;
  const bool h = (lane & 1) != 0;
  {
    const int rb = (int)(__brev((unsigned)b) >> 27);
    cft_half_pass(buf, rb, rb + 64, rb + 32, rb + 96, 4 * b + (h ? 1 : 0), 4 * b + (h ? 3 : 2), h,
                  L.tw0, (L.diag & 1) != 0);
  }
  wave_lds_fence();
  {
    const int base = 16 * (b >> 2) + (b & 3);
    cft_half_pass(buf, base, base + 4, base + 8, base + 12, base + (h ? 4 : 0),
                  base + (h ? 12 : 8), h, L.tw1, (L.diag & 2) != 0);
  }
  wave_lds_fence();
  {
    const int base = 64 * (b >> 4) + (b & 15);
    cft_half_pass(buf, base, base + 16, base + 32, base + 48, base + (h ? 16 : 0),
                  base + (h ? 48 : 32), h, L.tw2, (L.diag & 4) != 0);
  }
  wave_lds_fence();
}

// WebRtc_rdft(256, +1): buf holds the 128 complex inputs (x[2n], x[2n+1]) in
// natural order.  Returns complex elements q (lo) and q+64 (hi) of the Ooura
// packed spectrum; lane 0: lo = (R0, R128).
__device__ __forceinline__ void rdft256_fwd(float2* buf, const FftLane& L, int lane, float2& lo,
                                            float2& hi) {
  cft128_passes(buf, L, lane);
  const float2 a = buf[lane], c = buf[lane + 64];
  const float lor = a.x + c.x, loi = a.y + c.y;  // fft4g.c:939-947
  const float hir = a.x - c.x, hii = a.y - c.y;
  const int src = (64 - lane) & 63;
  const float plor = __shfl(lor, src, 64), ploi = __shfl(loi, src, 64);
  const float phir = __shfl(hir, src, 64), phii = __shfl(hii, src, 64);
  // rftfsub (fft4g.c:1234-1256): pair (j = q, k = 128 - q) updates lo ...
  const float wkr1 = 0.5f - L.cr, wki1 = L.cq;
  float xr = lor - phir, xi = loi + phii;
  float yr = wkr1 * xr - wki1 * xi, yi = wkr1 * xi + wki1 * xr;
  float nlor = lor - yr, nloi = loi - yi;
  // ... and pair (j = 64 - q, k = 64 + q) updates hi.
  const float wkr2 = 0.5f - L.cq, wki2 = L.cr;
  xr = plor - hir;
  xi = ploi + hii;
  yr = wkr2 * xr - wki2 * xi;
  yi = wkr2 * xi + wki2 * xr;
  float nhir = hir + yr, nhii = hii - yi;
  if (lane == 0) {  // fft4g.c:347-349, element 64 untouched
    nlor = lor + loi;
    nloi = lor - loi;
    nhir = hir;
    nhii = hii;
  }
  lo = make_float2(nlor, nloi);
  hi = make_float2(nhir, nhii);
}

// WebRtc_rdft(256, -1), unscaled.  In: packed spectrum elements q / q+64.
// Out: lo = (x[2q], x[2q+1]), hi = (x[2q+128], x[2q+129]).
__device__ __forceinline__ void rdft256_inv(float2* buf, const FftLane& L, int lane, float2& lo,
                                            float2& hi) {
  const float lor = lo.x, loi = lo.y, hir = hi.x, hii = hi.y;
  const int src = (64 - lane) & 63;
  const float plor = __shfl(lor, src, 64), ploi = __shfl(loi, src, 64);
  const float phir = __shfl(hir, src, 64), phii = __shfl(hii, src, 64);
  // rftbsub (fft4g.c:1259-1283)
  const float wkr1 = 0.5f - L.cr, wki1 = L.cq;
  float xr = lor - phir, xi = loi + phii;
  float yr = wkr1 * xr + wki1 * xi, yi = wkr1 * xi - wki1 * xr;
  float nlor = lor - yr, nloi = yi - loi;
  const float wkr2 = 0.5f - L.cq, wki2 = L.cr;
  xr = plor - hir;
  xi = ploi + hii;
  yr = wkr2 * xr + wki2 * xi;
  yi = wkr2 * xi - wki2 * xr;
  float nhir = hir + yr, nhii = yi - hii;
  if (lane == 0) {  // fft4g.c:351-352, :1264, :1282
    const float t = 0.5f * (lor - loi);
    nlor = lor - t;
    nloi = -t;
    nhir = hir;
    nhii = -hii;
  }
  buf[lane] = make_float2(nlor, nloi);
  buf[lane + 64] = make_float2(nhir, nhii);
  wave_lds_fence();
  cft128_passes(buf, L, lane);
  const float2 a = buf[lane], c = buf[lane + 64];
  lo = make_float2(a.x + c.x, -a.y - c.y);  // fft4g.c:989-997
  hi = make_float2(a.x - c.x, -a.y + c.y);
}

// ---- 8 kHz: WebRtc_rdft(128, +-1) = a 64-point complex transform (fft4g.c:902-937, 952-987: cft1st,
// cftmdl(l = 8), then ONE twiddle-free radix-4 stage since 4 l == n) + rftfsub / rftbsub with nc = 32.
// Lane = half butterfly as above; 16 butterflies per pass, so lanes 32..63 repeat the work of lanes
// 0..31 (same addresses, same values).  Element q of the result sits on lane q.
struct FftLane8 {
  float4 tw0, tw1;  // passes 1 and 2; pass 3 has no twiddles
  int diag;
  float ca, cb;     // makect(32): c[p], c[32 - p] for p = min(lane, 64 - lane)
};
__device__ __forceinline__ FftLane8 load_fft_lane8(const NsTables* __restrict__ T, int lane) {
  FftLane8 L;
  L.tw0 = *reinterpret_cast<const float4*>(T->tw8[0][lane]);
  L.tw1 = *reinterpret_cast<const float4*>(T->tw8[1][lane]);
  L.diag = T->diag8[lane];
  L.ca = T->c8a[lane];
  L.cb = T->c8b[lane];
  return L;
}
__device__ __forceinline__ void cft64_passes(float2* buf, const FftLane8& L, int lane, bool backward) {
  const int b = (lane >> 1) & 15;
  const bool h = (lane & 1) != 0;
  {
    const int rb = (int)(__brev((unsigned)b) >> 28);  // bitrv2 of 64 complex points
    cft_half_pass(buf, rb, rb + 32, rb + 16, rb + 48, 4 * b + (h ? 1 : 0), 4 * b + (h ? 3 : 2), h,
                  L.tw0, (L.diag & 1) != 0);
  }
  wave_lds_fence();
  {
    const int base = 16 * (b >> 2) + (b & 3);
    cft_half_pass(buf, base, base + 4, base + 8, base + 12, base + (h ? 4 : 0),
                  base + (h ? 12 : 8), h, L.tw1, (L.diag & 2) != 0);
  }
  wave_lds_fence();
  {
    // the last stage, twiddle-free: forward fft4g.c:918-937, backward :968-987 (which carries the
    // conjugation of the inverse transform).  With s = -1 on odd lanes (outputs 1 / 3) and +1 on even
    // ones (outputs 0 / 2), every line below is the reference's own sum or difference: x - y == x + (-y).
    const float2 c0 = buf[b], c1 = buf[b + 16], c2 = buf[b + 32], c3 = buf[b + 48];
    const uint32_t sm = h ? 0x80000000u : 0u;
    const float ur = c0.x + xorf(c1.x, sm);
    const float vr = c2.x + xorf(c3.x, sm), vi = c2.y + xorf(c3.y, sm);
    if (!backward) {
      const float ui = c0.y + xorf(c1.y, sm);
      const float vr2 = h ? -vi : vr;
      const float vi2 = h ? vr : vi;
      buf[b + (h ? 16 : 0)] = make_float2(ur + vr2, ui + vi2);
      buf[b + (h ? 48 : 32)] = make_float2(ur - vr2, ui - vi2);
    } else {
      // x0i = -a[j+1] - a[j1+1] (even lanes), x1i = -a[j+1] + a[j1+1] (odd lanes)
      const float ui = -c0.y + xorf(c1.y, sm ^ 0x80000000u);
      // even: (x0r + x2r, x0i - x2i) -> j, (x0r - x2r, x0i + x2i) -> j2;
      // odd:  (x1r + x3i, x1i + x3r) -> j3, (x1r - x3i, x1i - x3r) -> j1
      const float A = h ? vi : vr;
      const float B = h ? vr : -vi;
      buf[b + (h ? 48 : 0)] = make_float2(ur + A, ui + B);
      buf[b + (h ? 16 : 32)] = make_float2(ur - A, ui - B);
    }
  }
  wave_lds_fence();
}
// WebRtc_rdft(128, +1): buf holds the 64 complex inputs (x[2n], x[2n+1]); returns element `lane` of the
// Ooura-packed spectrum (lane 0: (R0, R64)).
__device__ __forceinline__ float2 rdft128_fwd(float2* buf, const FftLane8& L, int lane) {
  cft64_passes(buf, L, lane, false);
  const float2 a = buf[lane], pa = buf[(64 - lane) & 63];
  const bool jside = lane < 32;  // rftfsub (fft4g.c:1234-1256): pair (j = p, k = 64 - p), p = 1..31
  const float2 J = jside ? a : pa, K = jside ? pa : a;
  const float wkr = 0.5f - L.cb, wki = L.ca;
  const float xr = J.x - K.x, xi = J.y + K.y;
  const float yr = wkr * xr - wki * xi, yi = wkr * xi + wki * xr;
  float2 r = make_float2(jside ? a.x - yr : a.x + yr, a.y - yi);
  if (lane == 32) r = a;                                       // element m / 2 is left alone
  if (lane == 0) r = make_float2(a.x + a.y, a.x - a.y);        // fft4g.c:347-349
  return r;
}
// WebRtc_rdft(128, -1), unscaled: in = element `lane` of the packed spectrum, out = (x[2 lane], x[2 lane + 1]).
__device__ __forceinline__ float2 rdft128_inv(float2* buf, const FftLane8& L, int lane, float2 a) {
  const int src = (64 - lane) & 63;
  const float2 pa = make_float2(__shfl(a.x, src, 64), __shfl(a.y, src, 64));
  const bool jside = lane < 32;  // rftbsub (fft4g.c:1259-1283)
  const float2 J = jside ? a : pa, K = jside ? pa : a;
  const float wkr = 0.5f - L.cb, wki = L.ca;
  const float xr = J.x - K.x, xi = J.y + K.y;
  const float yr = wkr * xr + wki * xi, yi = wkr * xi - wki * xr;
  float2 r = make_float2(jside ? a.x - yr : a.x + yr, yi - a.y);
  if (lane == 32) r = make_float2(a.x, -a.y);                  // a[m + 1] = -a[m + 1]
  if (lane == 0) {                                             // fft4g.c:351-352, :1264
    const float t = 0.5f * (a.x - a.y);
    r = make_float2(a.x - t, -t);
  }
  buf[lane] = r;
  wave_lds_fence();
  cft64_passes(buf, L, lane, true);
  return buf[lane];
}

// --------------------------------------------------------------------------
// One 10 ms frame of one stream per wave.
//   DO_A && DO_P : Analyze(frame) then Process(frame) on the same frame with
//                  the streams' state "paired" (analyzeBuf == dataBuf,
//                  magnPrevAnalyze == magnPrevProcess, noise == noisePrev at
//                  rest), which holds as long as a stream has only ever been
//                  driven through this fused step -- the loop body of
//                  test_ns_module.cpp:97-99.  One forward FFT serves both.
//   DO_A only    : WebRtcNs_AnalyzeCore.
//   DO_P only    : WebRtcNs_ProcessCore (one band).
// IO16: frames are int16 PCM in HBM (the WAV drivers' format): the int16 -> float-S16 load is
// value preserving (channel_buffer.cc:43-53) and the store applies FloatS16ToS16
// (audio_util.h:41-49) after the WEBRTC_SPL_SAT of ns_core.c:1357-1359.
__device__ __forceinline__ short float_s16_to_s16(float v) {
  const float kMaxRound = 32767 - 0.5f, kMinRound = -32768 + 0.5f;
  if (v > 0) return v >= kMaxRound ? (short)32767 : (short)(v + 0.5f);
  return v <= kMinRound ? (short)-32768 : (short)(v - 0.5f);
}
template <bool IO16>
__device__ __forceinline__ void store_pair(float* y, int idx, float a, float b) {
  if (IO16) {
    short2 v;
    v.x = float_s16_to_s16(a);
    v.y = float_s16_to_s16(b);
    *reinterpret_cast<short2*>(reinterpret_cast<short*>(y) + idx) = v;
  } else {
    *reinterpret_cast<float2*>(y + idx) = make_float2(a, b);
  }
}

// G8: the 8 kHz geometry (blockLen 80, anaLen 128, 65 bins: ns_core.c:89-98).  Lane q owns bin q, the
// "q + 64" slot is dead (it runs on constants and is neither loaded, summed nor stored) and the tail
// slot is bin 64; lane l owns samples 2l, 2l+1 of the 128-sample buffers.
template <bool DO_A, bool DO_P, bool IO16 = false, bool G8 = false>
__global__ __launch_bounds__(256, 4) void ns_frame_kernel(float* __restrict__ state,
                                                       int32_t* __restrict__ hist_all,
                                                       const NsTables* __restrict__ T,
                                                       const float* __restrict__ in,
                                                       float* __restrict__ out, int num_streams) {
  __shared__ float2 lds[4][128];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wv;
  if (stream >= num_streams) return;
  float* __restrict__ st = state + (size_t)stream * kStreamDwords;
  float* __restrict__ vec = st + kOffVec;
  int32_t* __restrict__ hist = hist_all + (size_t)stream * kHistDwords;
  float2* buf = lds[wv];
  constexpr int BL = G8 ? 80 : kBlockL;      // samples per frame
  constexpr int AN = G8 ? 128 : kAnal;       // analysis window
  constexpr int NB = AN / 2 + 1;             // bins
  // x / NB, correctly rounded (DIV129's form for either bin count)
#define DIVB(a) div_by_uniform((a), (float)NB, 1.0f / (float)NB)
  // the lane's partial of a cross-bin sum: its q slot, plus its q + 64 slot where that exists
#define P2(a0, a1) (G8 ? (a0) : (a0) + (a1))

  // ---- per-stream scalars: lane k holds scalar k
  float sv = st[kOffScalars + lane];
#define SC_I(k) __builtin_amdgcn_readlane(__float_as_int(sv), (k))
#define SC_F(k) __int_as_float(SC_I(k))
#define SC_SET_I(k, val) sv = (lane == (k)) ? __int_as_float(val) : sv
#define SC_SET_F(k, val) sv = (lane == (k)) ? (val) : sv

  // ---- sliding analysis buffer: [96 carried samples | 160 new], lane l owns 4l..4l+3
  float* hbuf = st + ((DO_A) ? kOffAnaHist : kOffDataHist);
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), w4 = s4;
  if (G8) {
    // [48 carried | 80 new]: lanes 0..23 the carried samples, lanes 24..63 the frame (s4.z / .w unused)
    if (IO16) {
      const short* in16 = reinterpret_cast<const short*>(in) + (size_t)stream * BL;
      if (lane < 24) {
        const float2 c2 = *reinterpret_cast<const float2*>(hbuf + 2 * lane);
        s4.x = c2.x;
        s4.y = c2.y;
      } else {
        const short2 q = *reinterpret_cast<const short2*>(in16 + 2 * (lane - 24));
        s4.x = (float)q.x;
        s4.y = (float)q.y;
      }
    } else {
      const float* src = lane < 24 ? hbuf + 2 * lane : in + (size_t)stream * BL + 2 * (lane - 24);
      const float2 c2 = *reinterpret_cast<const float2*>(src);
      s4.x = c2.x;
      s4.y = c2.y;
    }
    const float2 w2 = *reinterpret_cast<const float2*>(T->window8 + 2 * lane);
    w4.x = w2.x;
    w4.y = w2.y;
  } else if (IO16) {
    const short* in16 = reinterpret_cast<const short*>(in) + (size_t)stream * kBlockL;
    if (lane < 24) {
      s4 = *reinterpret_cast<const float4*>(hbuf + 4 * lane);
    } else {
      const short4 q = *reinterpret_cast<const short4*>(in16 + 4 * (lane - 24));
      s4 = make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
    }
  } else {
    const float* src =
        lane < 24 ? hbuf + 4 * lane : in + (size_t)stream * kBlockL + 4 * (lane - 24);
    s4 = *reinterpret_cast<const float4*>(src);
  }
  if (!G8) w4 = *reinterpret_cast<const float4*>(T->window + 4 * lane);

  // ---- state rows (issued early; consumed after the FFT); bins lane and 64 + lane sit at
  // row_pos(lane) and 64 + row_pos(lane) (ns_layout.h)
  const int rpos = 4 * (lane & 15) + 2 * ((lane >> 4) & 1) + (lane >> 5);  // row_pos(lane)
  float LQ[3][3], DEN[3][3], quant[3], smooth[3], noisePrev[3], magnPrevA[3], logLrt[3],
      avgPause[3], noiseSt[3], magnPrevP[3];
#define LOAD_ROW(dst, f)                                          \
  {                                                               \
    dst[0] = vec[(f)*kVecStride + rpos];                          \
    dst[1] = G8 ? 1.f : vec[(f)*kVecStride + 64 + rpos];          \
    dst[2] = SC_F(S_TAIL0 + (f));                                 \
  }
#define STORE_ROW(f, srcv)                                        \
  {                                                               \
    vec[(f)*kVecStride + rpos] = srcv[0];                         \
    if (!G8) vec[(f)*kVecStride + 64 + rpos] = srcv[1];           \
    SC_SET_F(S_TAIL0 + (f), srcv[2]);                             \
  }
  if (DO_A) {
    LOAD_ROW(LQ[0], V_LQ0) LOAD_ROW(LQ[1], V_LQ1) LOAD_ROW(LQ[2], V_LQ2)
    LOAD_ROW(DEN[0], V_DEN0) LOAD_ROW(DEN[1], V_DEN1) LOAD_ROW(DEN[2], V_DEN2)
    LOAD_ROW(quant, V_QUANT)
  }
  float2 carry = make_float2(0.f, 0.f);  // syntBuf[0..95] (8 kHz: [0..47]), lane l owns 2l, 2l+1
  if (DO_P && lane < (AN - BL) / 2) carry = *reinterpret_cast<const float2*>(st + kOffSynt + 2 * lane);
  FftLane L;
  FftLane8 L8;
  if (G8) L8 = load_fft_lane8(T, lane); else L = load_fft_lane(T, lane);

  // Windowing + Energy (ns_core.c:969-978, 951-960)
  const float wx0 = w4.x * s4.x, wx1 = w4.y * s4.y, wx2 = w4.z * s4.z, wx3 = w4.w * s4.w;
  float epart = wx0 * wx0;
  epart += wx1 * wx1;
  if (!G8) {
    epart += wx2 * wx2;
    epart += wx3 * wx3;
  }
  const float energy1 = wave_sum(epart);

  int blockInd = SC_I(S_BLOCKIND);
  const float overdrive = SC_F(S_OVERDRIVE);
  const float denoiseBound = SC_F(S_DENOISEBOUND);
  float priorSpeechProb = SC_F(S_PRIORSPEECHPROB);

  // the carried 96 (48) samples for the next frame are this frame's last 96 (48)
  if (G8) {
    if (lane >= 40) *reinterpret_cast<float2*>(hbuf + 2 * (lane - 40)) = make_float2(s4.x, s4.y);
  } else if (lane >= 40) {
    *reinterpret_cast<float4*>(hbuf + 4 * (lane - 40)) = s4;
  }

  if (energy1 == 0.0f) {
    // Analyze: nothing but the buffer slide (ns_core.c:1072-1082).
    // Process: emit the synthesis tail (ns_core.c:1239-1264).
    if (DO_P) {
      float* sy = st + kOffSynt;
      float* y = IO16 ? reinterpret_cast<float*>(reinterpret_cast<short*>(out) + (size_t)stream * BL)
                      : out + (size_t)stream * BL;
      float2 o01 = carry;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      o01.x = o01.x > 32767 ? 32767 : (o01.x < -32768 ? -32768 : o01.x);
      o01.y = o01.y > 32767 ? 32767 : (o01.y < -32768 ? -32768 : o01.y);
      if (G8) {
        if (lane < 40) store_pair<IO16>(y, 2 * lane, o01.x, o01.y);  // the carry, then zeros
      } else {
        store_pair<IO16>(y, 2 * lane, o01.x, o01.y);
        if (lane < 16) store_pair<IO16>(y, 128 + 2 * lane, 0.f, 0.f);
      }
      if (lane < (AN - BL) / 2) *reinterpret_cast<float2*>(sy + 2 * lane) = make_float2(0.f, 0.f);
    }
    return;
  }

  // ---- forward FFT (ns_core.c:886-911)
  float2 lo, hi = make_float2(0.f, 0.f);
  if (G8) {
    buf[lane] = make_float2(wx0, wx1);
    wave_lds_fence();
    lo = rdft128_fwd(buf, L8, lane);
  } else {
    *reinterpret_cast<float4*>(&buf[2 * lane]) = make_float4(wx0, wx1, wx2, wx3);
    wave_lds_fence();
    rdft256_fwd(buf, L, lane, lo, hi);
  }
  // second group of state rows: issued after the FFT so they do not hold
  // registers across it; their latency hides under the magnitude / log / trackers
  asm volatile("" ::: "memory");
  if (DO_A) {
    LOAD_ROW(magnPrevA, V_MAGNPREV_A) LOAD_ROW(logLrt, V_LOGLRT) LOAD_ROW(avgPause, V_AVGPAUSE)
  }
  LOAD_ROW(smooth, V_SMOOTH) LOAD_ROW(noisePrev, V_NOISEPREV)
  if (DO_P && !DO_A) {
    LOAD_ROW(noiseSt, V_NOISE) LOAD_ROW(magnPrevP, V_MAGNPREV_P)
  }
  const float re128 = lane_bcast(lo.y, 0);
  float re[3], im[3], magn[3];
  re[0] = lo.x;
  im[0] = lane == 0 ? 0.f : lo.y;
  re[1] = hi.x;
  im[1] = hi.y;
  re[2] = re128;
  im[2] = 0.f;
  {
    const float m0 = fsqrt(re[0] * re[0] + im[0] * im[0]) + 1.f;
    magn[0] = lane == 0 ? fabsf(re[0]) + 1.f : m0;
    magn[1] = fsqrt(re[1] * re[1] + im[1] * im[1]) + 1.f;
    magn[2] = fabsf(re[2]) + 1.f;
  }

  float noise[3];      // noise estimate handed from Analyze to Process
  float prevStsaA[3];  // magnPrev / (noisePrev + 1e-4) * smooth, identical in both halves when paired

  if (DO_A) {
    blockInd++;  // ns_core.c:1084
    const int updateParsFlag = SC_I(S_MUP0);
    int updates = SC_I(S_UPDATES);
    int counter[3] = {SC_I(S_COUNTER0), SC_I(S_COUNTER1), SC_I(S_COUNTER2)};

    float lmagn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) lmagn[k] = log_f32_via_f64(magn[k]);

    // signalEnergy, sumMagn (ns_core.c:1088-1104)
    float t_se = P2(re[0] * re[0] + im[0] * im[0], re[1] * re[1] + im[1] * im[1]);
    float t_sm = P2(magn[0], magn[1]);
    if (lane == 0) {
      t_se = t_se + (re[2] * re[2] + im[2] * im[2]);
      t_sm = t_sm + magn[2];
    }
    float signalEnergy = wave_sum(t_se);
    const float sumMagn = wave_sum(t_sm);
    signalEnergy = DIVB(signalEnergy);

    // ---- NoiseEstimation (ns_core.c:217-285)
    if (updates < NS_END_STARTUP_LONG) updates++;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float cnt = (float)counter[s];
      const float cnt1 = (float)(counter[s] + 1);
      const float rcnt1 = 1.f / cnt1;  // correctly rounded, once per wave
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float den = DEN[s][k], lq = LQ[s][k];
        const float delta = den > 1.0f ? fdiv(NS_FACTOR * 1.f, den) : NS_FACTOR;
        const bool up = lmagn[k] > lq;
        const float step =
            div_by_uniform(up ? NS_QUANTILE * delta : (1.f - NS_QUANTILE) * delta, cnt1, rcnt1);
        lq = up ? lq + step : lq - step;
        const float nd = div_by_uniform(cnt * den + 1.f / (2.f * NS_WIDTH), cnt1, rcnt1);
        den = fabsf(lmagn[k] - lq) < NS_WIDTH ? nd : den;
        DEN[s][k] = den;
        LQ[s][k] = lq;
      }
      if (counter[s] >= NS_END_STARTUP_LONG) {
        counter[s] = 0;
        if (updates >= NS_END_STARTUP_LONG) {
#pragma unroll
          for (int k = 0; k < 3; ++k) quant[k] = exp_f32_via_f64(LQ[s][k], T->exp2_64);
        }
      }
      counter[s]++;
    }
    if (updates < NS_END_STARTUP_LONG) {
#pragma unroll
      for (int k = 0; k < 3; ++k) quant[k] = exp_f32_via_f64(LQ[2][k], T->exp2_64);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) noise[k] = quant[k];

    // ---- startup noise model (ns_core.c:1091-1100, 1109-1162)
    float whiteNoiseLevel = SC_F(S_WHITE);
    float pinkNoiseNumerator = SC_F(S_PINKNUM);
    float pinkNoiseExp = SC_F(S_PINKEXP);
    float fd5 = SC_F(S_FD5);
    if (blockInd < NS_END_STARTUP_SHORT) {
      float logi[3];
      logi[0] = T->logi[lane];
      logi[1] = G8 ? 0.f : T->logi[64 + lane];
      logi[2] = T->logi[NB - 1];
      float t_lm = P2(lane >= NS_START_BAND ? lmagn[0] : 0.f, lmagn[1]);
      float t_lilm = P2(lane >= NS_START_BAND ? logi[0] * lmagn[0] : 0.f, logi[1] * lmagn[1]);
      if (lane == 0) {
        t_lm = t_lm + lmagn[2];
        t_lilm = t_lilm + logi[2] * lmagn[2];
      }
      const float sum_log_magn = wave_sum(t_lm);
      const float sum_log_i_log_magn = wave_sum(t_lilm);
      const float sum_log_i = G8 ? T->sum_log_i8 : T->sum_log_i;
      const float sum_log_i_square = G8 ? T->sum_log_i_square8 : T->sum_log_i_square;
      whiteNoiseLevel += DIVB(sumMagn) * overdrive;
      float tmpFloat1 = sum_log_i_square * ((float)(NB - NS_START_BAND));
      tmpFloat1 -= (sum_log_i * sum_log_i);
      float tmpFloat2 = (sum_log_i_square * sum_log_magn - sum_log_i * sum_log_i_log_magn);
      float tmpFloat3 = tmpFloat2 / tmpFloat1;
      if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
      pinkNoiseNumerator += tmpFloat3;
      tmpFloat2 = (sum_log_i * sum_log_magn);
      tmpFloat2 -= ((float)(NB - NS_START_BAND)) * sum_log_i_log_magn;
      tmpFloat3 = tmpFloat2 / tmpFloat1;
      if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
      if (tmpFloat3 > 1.f) tmpFloat3 = 1.f;
      pinkNoiseExp += tmpFloat3;
      float parametric_num = 0.f, parametric_exp = 0.f;
      if (pinkNoiseExp > 0.f) {
        parametric_num = (float)exp((double)(pinkNoiseNumerator / (float)(blockInd + 1)));
        parametric_num *= (float)(blockInd + 1);
        parametric_exp = pinkNoiseExp / (float)(blockInd + 1);
      }
      float pn[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int bin = k == 0 ? lane : (k == 1 ? 64 + lane : NB - 1);
        if (pinkNoiseExp == 0.f) {
          pn[k] = whiteNoiseLevel;
        } else {
          const float use_band = (float)(bin < NS_START_BAND ? NS_START_BAND : bin);
          pn[k] = (float)((double)parametric_num / pow((double)use_band, (double)parametric_exp));
        }
        noise[k] *= (blockInd);
        const float t2 = pn[k] * (NS_END_STARTUP_SHORT - blockInd);
        noise[k] += (t2 / (float)(blockInd + 1));
        noise[k] /= NS_END_STARTUP_SHORT;
      }
      STORE_ROW(V_PARAMNOISE, pn)
    }
    if (blockInd < NS_END_STARTUP_LONG) {  // ns_core.c:1165-1169
      fd5 *= blockInd;
      fd5 += signalEnergy;
      fd5 /= (blockInd + 1);
    }

    // ---- ComputeSnr (ns_core.c:566-588)
    float snrLocPost[3], snrLocPrior[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float previousEstimateStsa = fdiv(magnPrevA[k], noisePrev[k] + 0.0001f) * smooth[k];
      prevStsaA[k] = previousEstimateStsa;
      snrLocPost[k] = 0.f;
      if (magn[k] > noise[k]) snrLocPost[k] = fdiv(magn[k], noise[k] + 0.0001f) - 1.f;
      snrLocPrior[k] =
          NS_DD_PR_SNR * previousEstimateStsa + (1.f - NS_DD_PR_SNR) * snrLocPost[k];
    }

    // ---- ComputeSpectralFlatness (ns_core.c:523-556)
    float fd0 = SC_F(S_FD0), fd4 = SC_F(S_FD4), fd6 = SC_F(S_FD6);
    float t_fl = P2(lane >= 1 ? lmagn[0] : 0.f, lmagn[1]);
    float t_ap = P2(avgPause[0], avgPause[1]);
    if (lane == 0) {
      t_fl = t_fl + lmagn[2];
      t_ap = t_ap + avgPause[2];
    }
    {
      float num = wave_sum(t_fl);
      float den = sumMagn - lane_bcast(magn[0], 0);
      den = DIVB(den);
      num = DIVB(num);
      const float spectralTmp = fdiv(exp_f32_via_f64(num, T->exp2_64), den);
      fd0 += NS_SPECT_FL_TAVG * (spectralTmp - fd0);
    }
    // ---- ComputeSpectralDifference (ns_core.c:595-634)
    {
      float avgPauseMean = wave_sum(t_ap);
      float avgMagn = sumMagn;
      avgPauseMean = DIVB(avgPauseMean);
      avgMagn = DIVB(avgMagn);
      float dm[3], dp[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dm[k] = magn[k] - avgMagn;
        dp[k] = avgPause[k] - avgPauseMean;
      }
      float t_cov = P2(dm[0] * dp[0], dm[1] * dp[1]);
      float t_vp = P2(dp[0] * dp[0], dp[1] * dp[1]);
      float t_vm = P2(dm[0] * dm[0], dm[1] * dm[1]);
      if (lane == 0) {
        t_cov = t_cov + dm[2] * dp[2];
        t_vp = t_vp + dp[2] * dp[2];
        t_vm = t_vm + dm[2] * dm[2];
      }
      float covMagnPause = wave_sum(t_cov);
      float varPause = wave_sum(t_vp);
      float varMagn = wave_sum(t_vm);
      covMagnPause = DIVB(covMagnPause);
      varPause = DIVB(varPause);
      varMagn = DIVB(varMagn);
      fd6 += signalEnergy;
      float avgDiffNormMagn = varMagn - fdiv(covMagnPause * covMagnPause, varPause + 0.0001f);
      avgDiffNormMagn = fdiv(avgDiffNormMagn, fd5 + 0.0001f);
      fd4 += NS_SPECT_DIFF_TAVG * (avgDiffNormMagn - fd4);
    }

    // ---- histograms / prior model (FeatureUpdate, ns_core.c:766-790)
    float fd3 = SC_F(S_FD3);  // previous frame's average LRT feeds the histogram
    PriorModel pm;
    pm.p0 = SC_F(S_PMP0);
    pm.p1 = SC_F(S_PMP1);
    pm.p3 = SC_F(S_PMP3);
    pm.p4 = SC_F(S_PMP4);
    pm.p5 = SC_F(S_PMP5);
    pm.p6 = SC_F(S_PMP6);
    const float pmp2 = SC_F(S_PMP2);
    int mup0 = updateParsFlag, mup3 = SC_I(S_MUP3);
    bool window_closed = false;
    const int mup1 = SC_I(S_MUP1);
    if (updateParsFlag >= 1) {
      mup3--;
      if (mup3 > 0) {  // FeatureParameterExtraction(self, 0), ns_core.c:309-334
        if (lane == 0) {
          if ((fd3 < kHist * 0.1f) && (fd3 >= 0.0f)) hist[(int)div_by_uniform(fd3, 0.1f, 1.0f / 0.1f)]++;
          if ((fd0 < kHist * 0.05f) && (fd0 >= 0.0f)) hist[kHistStride + (int)div_by_uniform(fd0, 0.05f, 1.0f / 0.05f)]++;
          if ((fd4 < kHist * 0.1f) && (fd4 >= 0.0f))
            hist[2 * kHistStride + (int)div_by_uniform(fd4, 0.1f, 1.0f / 0.1f)]++;
        }
      }
      if (mup3 == 0) {
        pm = close_histogram_window(hist, lane, mup1, mup0 >= 1, pm);
        window_closed = true;
        mup3 = mup1;
        if (updateParsFlag == 1) {
          mup0 = 0;
        } else {
          fd6 = fd6 / ((float)mup1);
          fd5 = 0.5f * (fd6 + fd5);
          fd6 = 0.f;
        }
      }
    }

    // ---- SpeechNoiseProb (ns_core.c:642-749)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float t1 = 1.f + 2.f * snrLocPrior[k];
      const float t2 = fdiv(2.f * snrLocPrior[k], t1 + 0.0001f);
      const float besselTmp = (snrLocPost[k] + 1.f) * t2;
      logLrt[k] += NS_LRT_TAVG * (besselTmp - log_f32_via_f64(t1) - logLrt[k]);
    }
    float t_ll = P2(logLrt[0], logLrt[1]);
    if (lane == 0) t_ll = t_ll + logLrt[2];
    float logLrtTimeAvgKsum = wave_sum(t_ll);
    logLrtTimeAvgKsum = DIVB(logLrtTimeAvgKsum);
    fd3 = logLrtTimeAvgKsum;
    {
      const float widthPrior0 = NS_WIDTH_PR_MAP, widthPrior1 = 2.f * NS_WIDTH_PR_MAP,
                  widthPrior2 = 2.f * NS_WIDTH_PR_MAP;
      const int sgnMap = (int)pmp2;
      float widthPrior = widthPrior0;
      if (logLrtTimeAvgKsum < pm.p0) widthPrior = widthPrior1;
      const float arg0 = widthPrior * (logLrtTimeAvgKsum - pm.p0);
      widthPrior = widthPrior0;
      if (sgnMap == 1 && (fd0 > pm.p1)) widthPrior = widthPrior1;
      if (sgnMap == -1 && (fd0 < pm.p1)) widthPrior = widthPrior1;
      const float arg1 = (float)sgnMap * widthPrior * (pm.p1 - fd0);
      widthPrior = widthPrior0;
      if (fd4 < pm.p3) widthPrior = widthPrior2;
      const float arg2 = widthPrior * (fd4 - pm.p3);
      // the three tanh() of :696-725 evaluated on lanes 0..2 of one call
      const float arg = lane == 0 ? arg0 : (lane == 1 ? arg1 : arg2);
      const float th = tanh_f32_via_f64(arg, T->exp2_64);
      const float indicator0 = 0.5f * (lane_bcast(th, 0) + 1.f);
      const float indicator1 = 0.5f * (lane_bcast(th, 1) + 1.f);
      const float indicator2 = 0.5f * (lane_bcast(th, 2) + 1.f);
      const float indPrior = pm.p4 * indicator0 + pm.p5 * indicator1 + pm.p6 * indicator2;
      priorSpeechProb += NS_PRIOR_UPDATE * (indPrior - priorSpeechProb);
      if (priorSpeechProb > 1.f) priorSpeechProb = 1.f;
      if (priorSpeechProb < 0.01f) priorSpeechProb = 0.01f;
    }
    float probSpeech[3];
    {
      const float gainPrior = fdiv(1.f - priorSpeechProb, priorSpeechProb + 0.0001f);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float invLrt = exp_f32_via_f64(-logLrt[k], T->exp2_64);
        invLrt = (float)gainPrior * invLrt;
        probSpeech[k] = fdiv(1.f, 1.f + invLrt);
      }
    }

    // ---- UpdateNoiseEstimate (ns_core.c:800-846).  The time constant carried
    // into bin i is the one bin i-1 selected from its speech probability.
    {
      const float upA = dpp_move<0x138>(probSpeech[0]);  // wave_shr:1 : lane q <- lane q-1
      float upB = dpp_move<0x138>(probSpeech[1]);
      const float a63 = lane_bcast(probSpeech[0], 63);
      const float b63 = lane_bcast(probSpeech[1], 63);
      if (lane == 0) upB = a63;
      float prevProb[3] = {upA, upB, G8 ? a63 : b63};  // the last bin follows bin 63 (8 kHz) / 127
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float gammaOld = prevProb[k] > NS_PROB_RANGE ? NS_SPEECH_UPDATE : NS_NOISE_UPDATE;
        if (k == 0 && lane == 0) gammaOld = NS_NOISE_UPDATE;
        const float ps = probSpeech[k], pns = 1.f - probSpeech[k];
        const float noiseUpdateTmp =
            gammaOld * noisePrev[k] +
            (1.f - gammaOld) * (pns * magn[k] + ps * noisePrev[k]);
        float gammaNew = NS_NOISE_UPDATE;
        if (ps > NS_PROB_RANGE) gammaNew = NS_SPEECH_UPDATE;
        if (ps < NS_PROB_RANGE) avgPause[k] += NS_GAMMA_PAUSE * (magn[k] - avgPause[k]);
        float nz;
        if (gammaNew == gammaOld) {
          nz = noiseUpdateTmp;
        } else {
          nz = gammaNew * noisePrev[k] +
               (1.f - gammaNew) * (pns * magn[k] + ps * noisePrev[k]);
          if (noiseUpdateTmp < nz) nz = noiseUpdateTmp;
        }
        noise[k] = nz;
      }
    }

    // ---- commit Analyze state
    STORE_ROW(V_LQ0, LQ[0]) STORE_ROW(V_LQ1, LQ[1]) STORE_ROW(V_LQ2, LQ[2])
    STORE_ROW(V_DEN0, DEN[0]) STORE_ROW(V_DEN1, DEN[1]) STORE_ROW(V_DEN2, DEN[2])
    STORE_ROW(V_QUANT, quant)
    STORE_ROW(V_LOGLRT, logLrt) STORE_ROW(V_AVGPAUSE, avgPause)
    STORE_ROW(V_MAGNPREV_A, magn)  // ns_core.c:1180
    if (!DO_P) STORE_ROW(V_NOISE, noise)  // ns_core.c:1179
    SC_SET_I(S_UPDATES, updates);
    SC_SET_I(S_COUNTER0, counter[0]);
    SC_SET_I(S_COUNTER1, counter[1]);
    SC_SET_I(S_COUNTER2, counter[2]);
    SC_SET_I(S_MUP0, mup0);
    SC_SET_I(S_MUP3, mup3);
    SC_SET_F(S_SIGNALENERGY, signalEnergy);
    SC_SET_F(S_SUMMAGN, sumMagn);
    if (blockInd < NS_END_STARTUP_SHORT) {  // only the start-up model moves these
      SC_SET_F(S_WHITE, whiteNoiseLevel);
      SC_SET_F(S_PINKNUM, pinkNoiseNumerator);
      SC_SET_F(S_PINKEXP, pinkNoiseExp);
    }
    if (window_closed) {  // only FeatureParameterExtraction(self, 1) moves these
      SC_SET_F(S_PMP0, pm.p0);
      SC_SET_F(S_PMP1, pm.p1);
      SC_SET_F(S_PMP3, pm.p3);
      SC_SET_F(S_PMP4, pm.p4);
      SC_SET_F(S_PMP5, pm.p5);
      SC_SET_F(S_PMP6, pm.p6);
    }
    SC_SET_F(S_FD0, fd0);
    SC_SET_F(S_FD3, fd3);
    SC_SET_F(S_FD4, fd4);
    SC_SET_F(S_FD5, fd5);
    SC_SET_F(S_FD6, fd6);
    SC_SET_I(S_BLOCKIND, blockInd);
    SC_SET_F(S_PRIORSPEECHPROB, priorSpeechProb);
  }

  if (DO_P) {
    if (!DO_A) {
#pragma unroll
      for (int k = 0; k < 3; ++k) noise[k] = noiseSt[k];
    }
    float mprev[3];  // paired: magnPrevProcess == magnPrevAnalyze (previous frame's magn)
#pragma unroll
    for (int k = 0; k < 3; ++k) mprev[k] = DO_A ? magnPrevA[k] : magnPrevP[k];
    const int gainmap = SC_I(S_GAINMAP);

    float initMagn[3], pnoise[3];
    if (blockInd < NS_END_STARTUP_SHORT) {  // ns_core.c:1268-1272
      LOAD_ROW(initMagn, V_INITMAGN)
      LOAD_ROW(pnoise, V_PARAMNOISE)
#pragma unroll
      for (int k = 0; k < 3; ++k) initMagn[k] += magn[k];
      STORE_ROW(V_INITMAGN, initMagn)
    }
    float gainv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // ComputeDdBasedWienerFilter (ns_core.c:985-1007)
      const float previousEstimateStsa =
          DO_A ? prevStsaA[k] : fdiv(mprev[k], noisePrev[k] + 0.0001f) * smooth[k];
      float currentEstimateStsa = 0.f;
      if (magn[k] > noise[k]) currentEstimateStsa = fdiv(magn[k], noise[k] + 0.0001f) - 1.f;
      const float snrPrior =
          NS_DD_PR_SNR * previousEstimateStsa + (1.f - NS_DD_PR_SNR) * currentEstimateStsa;
      float g = fdiv(snrPrior, overdrive + snrPrior);
      // floors and startup blend (ns_core.c:1276-1307)
      if (g < denoiseBound) g = denoiseBound;
      if (g > 1.f) g = 1.f;
      if (blockInd < NS_END_STARTUP_SHORT) {
        float tmp = (initMagn[k] - overdrive * pnoise[k]);
        tmp /= (initMagn[k] + 0.0001f);
        if (tmp < denoiseBound) tmp = denoiseBound;
        if (tmp > 1.f) tmp = 1.f;
        g *= (blockInd);
        tmp *= (NS_END_STARTUP_SHORT - blockInd);
        g += tmp;
        g /= (NS_END_STARTUP_SHORT);
      }
      gainv[k] = g;
      re[k] *= g;
      im[k] *= g;
    }
    STORE_ROW(V_SMOOTH, gainv)           // ns_core.c:1304
    STORE_ROW(V_NOISEPREV, noise)        // ns_core.c:1310
    if (!DO_A) STORE_ROW(V_MAGNPREV_P, magn)  // ns_core.c:1309 (paired: V_MAGNPREV_A holds it)

    // ---- IFFT (ns_core.c:923-944)
    float2 tlo = make_float2(re[0], lane == 0 ? re[2] : im[0]);
    float2 thi = make_float2(re[1], im[1]);
    if (G8) {
      tlo = rdft128_inv(buf, L8, lane, tlo);
    } else {
      int lane_o = lane;  // opaque copy: keeps the compiler from holding the forward tables live
      asm volatile("" : "+v"(lane_o));
      const FftLane Li = load_fft_lane(T, lane_o);
      rdft256_inv(buf, Li, lane, tlo, thi);
    }
    float td0 = tlo.x * (2.f / AN), td1 = tlo.y * (2.f / AN);
    float td2 = thi.x * (2.f / AN), td3 = thi.y * (2.f / AN);

    // ---- energy-based gain compensation (ns_core.c:1315-1342)
    float factor = 1.f;
    if (gainmap == 1 && blockInd > NS_END_STARTUP_LONG) {
      float factor1 = 1.f, factor2 = 1.f;
      float e2 = td0 * td0;
      e2 += td1 * td1;
      if (!G8) {
        e2 += td2 * td2;
        e2 += td3 * td3;
      }
      const float energy2 = wave_sum(e2);
      float gain = fsqrt(fdiv(energy2, energy1 + 1.f));
      if (gain > NS_B_LIM) {
        factor1 = 1.f + 1.3f * (gain - NS_B_LIM);
        if (gain * factor1 > 1.f) factor1 = fdiv(1.f, gain);
      }
      if (gain < NS_B_LIM) {
        if (gain <= denoiseBound) gain = denoiseBound;
        factor2 = 1.f - 0.3f * (NS_B_LIM - gain);
      }
      factor = priorSpeechProb * factor1 + (1.f - priorSpeechProb) * factor2;
    }

    // ---- synthesis window, overlap-add, emit 160, carry 96 (ns_core.c:1344-1359)
    const float2 wlo = *reinterpret_cast<const float2*>((G8 ? T->window8 : T->window) + 2 * lane);
    float* sy = st + kOffSynt;
    float* y = IO16 ? reinterpret_cast<float*>(reinterpret_cast<short*>(out) + (size_t)stream * BL)
                    : out + (size_t)stream * BL;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // carry was read long ago; keep it so
    float o0 = carry.x + factor * (wlo.x * td0);
    float o1 = carry.y + factor * (wlo.y * td1);
    if (G8) {
      // samples 0..79 leave, samples 80..127 (lanes 40..63) become the next carry
      if (lane >= 40) *reinterpret_cast<float2*>(sy + 2 * (lane - 40)) = make_float2(o0, o1);
      o0 = o0 > 32767 ? 32767 : (o0 < -32768 ? -32768 : o0);
      o1 = o1 > 32767 ? 32767 : (o1 < -32768 ? -32768 : o1);
      if (lane < 40) store_pair<IO16>(y, 2 * lane, o0, o1);
    } else {
      const float2 whi = *reinterpret_cast<const float2*>(T->window + 128 + 2 * lane);
      float o2 = 0.f + factor * (whi.x * td2);
      float o3 = 0.f + factor * (whi.y * td3);
      if (lane >= 16) {  // samples 160..255 become the next carry
        *reinterpret_cast<float2*>(sy + 2 * lane - 32) = make_float2(o2, o3);
      }
      o0 = o0 > 32767 ? 32767 : (o0 < -32768 ? -32768 : o0);
      o1 = o1 > 32767 ? 32767 : (o1 < -32768 ? -32768 : o1);
      store_pair<IO16>(y, 2 * lane, o0, o1);
      if (lane < 16) {
        o2 = o2 > 32767 ? 32767 : (o2 < -32768 ? -32768 : o2);
        o3 = o3 > 32767 ? 32767 : (o3 < -32768 ? -32768 : o3);
        store_pair<IO16>(y, 128 + 2 * lane, o2, o3);
      }
    }
  }

  st[kOffScalars + lane] = sv;  // Process alone changes no scalar but the rows' bin-128 slots
#undef SC_I
#undef SC_F
#undef SC_SET_I
#undef SC_SET_F
#undef LOAD_ROW
#undef STORE_ROW
#undef DIVB
#undef P2
}

// Leaves the paired representation: materialises dataBuf / magnPrevProcess /
// noise copies so Analyze and Process can be called separately afterwards.
__global__ __launch_bounds__(256) void ns_unpair_kernel(float* __restrict__ state,
                                                        int num_streams) {
  const int lane = threadIdx.x & 63;
  const int stream = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (stream >= num_streams) return;
  float* st = state + (size_t)stream * kStreamDwords;
  float* vec = st + kOffVec;
  for (int i = lane; i < kCarry; i += 64) st[kOffDataHist + i] = st[kOffAnaHist + i];
  for (int i = lane; i < kVecStride; i += 64) {
    vec[V_MAGNPREV_P * kVecStride + i] = vec[V_MAGNPREV_A * kVecStride + i];
    vec[V_NOISE * kVecStride + i] = vec[V_NOISEPREV * kVecStride + i];
  }
  if (lane == 0) {  // bin 128 of the two rows (kept with the scalars)
    st[kOffScalars + S_TAIL0 + V_MAGNPREV_P] = st[kOffScalars + S_TAIL0 + V_MAGNPREV_A];
    st[kOffScalars + S_TAIL0 + V_NOISE] = st[kOffScalars + S_TAIL0 + V_NOISEPREV];
  }
}

// WebRtcNs_set_policy_core (ns_core.c:1013-1041) for every stream.
__global__ void ns_set_policy_kernel(float* __restrict__ state, int num_streams, int mode,
                                     float overdrive, float denoiseBound, int gainmap) {
  const int stream = blockIdx.x * blockDim.x + threadIdx.x;
  if (stream >= num_streams) return;
  float* sc = state + (size_t)stream * kStreamDwords + kOffScalars;
  sc[S_AGGRMODE] = __int_as_float(mode);
  sc[S_OVERDRIVE] = overdrive;
  sc[S_DENOISEBOUND] = denoiseBound;
  sc[S_GAINMAP] = __int_as_float(gainmap);
}

// Test seams for the device math: fn 0 = log_f32_via_f64 (lean path + fallback),
// 1 = (float)log((double)x), 2 = (float)exp((double)x), 3 = (float)tanh((double)x),
// 4..8 division forms, 9/10 sqrtf / fsqrt, 11/12 the kernels' exp / tanh.
__device__ __forceinline__ float debug_fn(int fn, float x, float param,
                                          const double* __restrict__ t64,
                                          const double2* __restrict__ logtab) {
  switch (fn) {
    case 19: return log_f32_via_tab(x, logtab);
    // batched forms (two arguments: x and the fixed `param`, whose result must equal the
    // one-argument form's; a wrong second result poisons the first)
    case 20: {
      const float in[2] = {x, param};
      float o[2];
      log_f32_via_tab_n<2>(in, o, logtab);
      return __float_as_uint(o[1]) == __float_as_uint(log_f32_via_tab(param, logtab)) ? o[0] : __builtin_nanf("");
    }
    case 21: {
      const float in[2] = {x, param};
      float o[2];
      exp_f32_via_f64_n<2>(in, o, t64);
      return __float_as_uint(o[1]) == __float_as_uint(exp_f32_via_f64(param, t64)) ? o[0] : __builtin_nanf("");
    }
    case 22: {
      const float in[2] = {x, param};
      float o[2];
      fsqrt_n<2>(in, o);
      return __float_as_uint(o[1]) == __float_as_uint(fsqrt(param)) ? o[0] : __builtin_nanf("");
    }
    case 9: return sqrtf(x);
    case 10: return fsqrt(x);
    case 11: return exp_f32_via_f64(x, t64);
    case 12: return tanh_f32_via_f64(x, t64);
    case 4: return x / param;
    case 5: return div_by_uniform(x, param, 1.0f / param);
    case 6: return fdiv(x, param);
    case 7: return param / x;
    case 8: return fdiv(param, x);
    case 0: return log_f32_via_f64(x);
    case 1: return (float)log((double)x);
    case 2: return (float)exp((double)x);
    case 3: return (float)tanh((double)x);
    case 13: return (float)pow((double)x, (double)param);
    case 14: return pow_f32_via_f64(x, param, t64);
    case 15: return (float)cos((double)x);
    case 16: { float sv, cv; sincos_f32_via_f64(x, sv, cv); return cv; }
    case 17: return (float)sin((double)x);
    case 18: { float sv, cv; sincos_f32_via_f64(x, sv, cv); return sv; }
    default: return x;
  }
}
__global__ void debug_eval_kernel(int fn, float* data, size_t n,
                                  const NsTables* __restrict__ T) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) data[i] = debug_fn(fn, data[i], 1.0f, T->exp2_64, reinterpret_cast<const double2*>(T->logtab));
}
// Compares fn_a and fn_b on every float whose bit pattern is in [start, start+count).
__global__ void debug_compare_kernel(int fn_a, int fn_b, unsigned start, unsigned count,
                                     unsigned* n_bad, unsigned* bad_bits, float param,
                                     const NsTables* __restrict__ T) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float x = __uint_as_float(start + i);
  const double2* lt = reinterpret_cast<const double2*>(T->logtab);
  const float a = debug_fn(fn_a, x, param, T->exp2_64, lt), b = debug_fn(fn_b, x, param, T->exp2_64, lt);
  const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
  if (!same) {
    const unsigned k = atomicAdd(n_bad, 1u);
    if (k < 64) bad_bits[k] = start + i;
  }
}

// FFT seam for the parity tests: WebRtc_rdft(256, isgn) on each 256-float row.
__global__ __launch_bounds__(256) void rdft256_kernel(float* __restrict__ data, int count,
                                                      int isgn,
                                                      const NsTables* __restrict__ T) {
  __shared__ float2 lds[4][128];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = blockIdx.x * 4 + wv;
  if (row >= count) return;
  float* a = data + (size_t)row * kAnal;
  float2* buf = lds[wv];
  const FftLane L = load_fft_lane(T, lane);
  float2 lo, hi;
  if (isgn >= 0) {
    *reinterpret_cast<float4*>(&buf[2 * lane]) = *reinterpret_cast<const float4*>(a + 4 * lane);
    wave_lds_fence();
    rdft256_fwd(buf, L, lane, lo, hi);
  } else {
    lo = *reinterpret_cast<const float2*>(a + 2 * lane);
    hi = *reinterpret_cast<const float2*>(a + 128 + 2 * lane);
    rdft256_inv(buf, L, lane, lo, hi);
  }
  *reinterpret_cast<float2*>(a + 2 * lane) = lo;
  *reinterpret_cast<float2*>(a + 128 + 2 * lane) = hi;
}
// The same seam for WebRtc_rdft(128, isgn) (the 8 kHz transform) on each 128-float row.
__global__ __launch_bounds__(256) void rdft128_kernel(float* __restrict__ data, int count,
                                                      int isgn,
                                                      const NsTables* __restrict__ T) {
  __shared__ float2 lds[4][128];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = blockIdx.x * 4 + wv;
  if (row >= count) return;
  float* a = data + (size_t)row * 128;
  float2* buf = lds[wv];
  const FftLane8 L = load_fft_lane8(T, lane);
  float2 v = *reinterpret_cast<const float2*>(a + 2 * lane);
  if (isgn >= 0) {
    buf[lane] = v;
    wave_lds_fence();
    v = rdft128_fwd(buf, L, lane);
  } else {
    v = rdft128_inv(buf, L, lane, v);
  }
  *reinterpret_cast<float2*>(a + 2 * lane) = v;
}

}  // namespace

// ------------------------------------------------------------ launch wrappers
namespace aspns {

// mode: 0 Analyze, 1 Process, 2 fused, 3 fused with int16 frames; g8: the 8 kHz geometry
hipError_t launch_ns_frame(int mode, float* state, int32_t* hist, const NsTables* T,
                           const float* in, float* out, int num_streams, hipStream_t s, bool g8) {
  const dim3 grid((num_streams + 3) / 4), block(256);
#define NS_LAUNCH(A, P, I16, G)                                                                   \
  hipLaunchKernelGGL((ns_frame_kernel<A, P, I16, G>), grid, block, 0, s, state, hist, T, in, out, \
                     num_streams)
  if (g8) {
    switch (mode) {
      case 0: NS_LAUNCH(true, false, false, true); break;
      case 1: NS_LAUNCH(false, true, false, true); break;
      case 3: NS_LAUNCH(true, true, true, true); break;
      default: NS_LAUNCH(true, true, false, true); break;
    }
  } else {
    switch (mode) {
      case 0: NS_LAUNCH(true, false, false, false); break;
      case 1: NS_LAUNCH(false, true, false, false); break;
      case 3: NS_LAUNCH(true, true, true, false); break;
      default: NS_LAUNCH(true, true, false, false); break;
    }
  }
#undef NS_LAUNCH
  return hipGetLastError();
}

hipError_t launch_ns_unpair(float* state, int num_streams, hipStream_t s) {
  hipLaunchKernelGGL(ns_unpair_kernel, dim3((num_streams + 3) / 4), dim3(256), 0, s, state,
                     num_streams);
  return hipGetLastError();
}

hipError_t launch_ns_set_policy(float* state, int num_streams, int mode, float overdrive,
                                float denoiseBound, int gainmap, hipStream_t s) {
  hipLaunchKernelGGL(ns_set_policy_kernel, dim3((num_streams + 255) / 256), dim3(256), 0, s,
                     state, num_streams, mode, overdrive, denoiseBound, gainmap);
  return hipGetLastError();
}

hipError_t launch_debug_eval(int fn, float* data, size_t n, const NsTables* T, hipStream_t s) {
  hipLaunchKernelGGL(debug_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn,
                     data, n, T);
  return hipGetLastError();
}
hipError_t launch_debug_compare(int fn_a, int fn_b, unsigned start, unsigned count,
                                unsigned* n_bad, unsigned* bad_bits, float param,
                                const NsTables* T, hipStream_t s) {
  hipLaunchKernelGGL(debug_compare_kernel, dim3((count + 255) / 256), dim3(256), 0, s, fn_a, fn_b,
                     start, count, n_bad, bad_bits, param, T);
  return hipGetLastError();
}

hipError_t launch_rdft256(float* data, int count, int isgn, const NsTables* T, hipStream_t s, int n) {
  if (n == 128)
    hipLaunchKernelGGL(rdft128_kernel, dim3((count + 3) / 4), dim3(256), 0, s, data, count, isgn, T);
  else
    hipLaunchKernelGGL(rdft256_kernel, dim3((count + 3) / 4), dim3(256), 0, s, data, count, isgn, T);
  return hipGetLastError();
}

}  // namespace aspns
