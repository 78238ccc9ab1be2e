// ns_kernels2.hip -- the fused Analyze+Process frame step with TWO streams per wave64.
//
// Same arithmetic as ns_frame_kernel<true,true> (ns_kernels.hip) -- every per-bin float
// operation of ns_core.c:1043-1359 in the reference's order, Ooura-order FFT, exact libm forms
// -- re-mapped for instruction issue, which is what bounds the one-stream-per-wave kernel
// (profiles/README.md: ~4 cycles per wave64 VALU instruction, no cross-wave overlap):
//
//   * lanes 0-31 carry stream 2w, lanes 32-63 stream 2w+1; a lane owns 4 bins
//     (lane L < 16: bins L, L+16, L+32, L+48; lane 16+L: 64+L, 80+L, 96+L, 112+L) plus, on lane
//     0 of each half, bin 128: 5 slot passes serve two streams (2.5 per stream instead of 3);
//   * every lane computes one FULL radix-4 butterfly per FFT pass (32 butterflies per stream and
//     pass), data moving between passes through a 1 KB LDS tile per stream, the radix-2 tail
//     through one ds_swizzle, the real split through one LDS gather -- no duplicated work;
//   * the ~10 cross-bin sums per frame are 32-lane butterflies (4 DPP steps + 1 swizzle), one
//     instruction stream reducing both streams at once; the per-stream scalar section (feature
//     updates, tanh, prior model, histogram) likewise runs once per wave for two streams;
//   * state stays in the natural-bin-order layout of ns_layout.h (no conversion between the two
//     kernels): a lane's 4 bins are 64 B apart, so a load instruction touches four 64-byte
//     segments -- the same bytes, 3x fewer memory instructions per stream.
//
// The sums' association is the 32-lane one (lane-local over its 4 bins, then xor 1,2,4,8,16, then bin 128),
// which oracle/ns_oracle.c reproduces as ASP_NS_REDUCE_TREE32; the tests compare bit for bit.
// Needs an even stream count (the host sends an odd last stream to the one-stream kernel).
#include <hip/hip_runtime.h>

#include "ns_device.h"
#include "ns_layout.h"

namespace {
using namespace aspns_dev;

constexpr int NS5 = 5;  // 4 owned bins + the tail bin 128

__device__ __forceinline__ float swz_xor16(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
// value of `v` on lane k of the caller's own half-wave
__device__ __forceinline__ float half_lane(float v, int hb, int k) { return __shfl(v, hb + k, 64); }

// 32-lane all-reduce, ascending xor butterfly (1, 2, 4, 8, 16)
__device__ __forceinline__ float half_sum(float v) {
  v = v + dpp_move<0xB1>(v);
  v = v + dpp_move<0x4E>(v);
  v = v + dpp_move<0x141>(v);
  v = v + dpp_move<0x140>(v);
  // xor 16: v_permlane16_swap (gfx950) trades rows 1 <-> 0 and 3 <-> 2 of two copies, so one copy
  // ends up holding the even row's value on both rows of a pair and the other the odd row's; their
  // sum is own + partner on every lane (the addition commutes) without a trip through the LDS pipe
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One full radix-4 butterfly of cft1st / cftmdl (fft4g.c:1002-1231); tw = (w1r, w1i, w2r, w2i,
// w3r, w3i, diag, -) from the per-lane table (identity for twiddle-free blocks, w2 = (0, 1) and
// the factored w[2] forms selected by `diag` for the second block of a pass).
__device__ __forceinline__ void bfly4(const float2 c[4], const float* __restrict__ tw,
                                      float2 o[4]) {
  const float4 ta = *reinterpret_cast<const float4*>(tw);
  const float4 tb = *reinterpret_cast<const float4*>(tw + 4);
  const bool diag = tb.z != 0.0f;
  const float x0r = c[0].x + c[1].x, x0i = c[0].y + c[1].y;
  const float x1r = c[0].x - c[1].x, x1i = c[0].y - c[1].y;
  const float x2r = c[2].x + c[3].x, x2i = c[2].y + c[3].y;
  const float x3r = c[2].x - c[3].x, x3i = c[2].y - c[3].y;
  o[0] = make_float2(x0r + x2r, x0i + x2i);
  const float dr = x0r - x2r, di = x0i - x2i;
  o[2] = make_float2(ta.z * dr - ta.w * di, ta.z * di + ta.w * dr);
  const float yr = x1r - x3i, yi = x1i + x3r;
  const float zr = x1r + x3i, zi = x1i - x3r;
  const float g1r = ta.x * yr - ta.y * yi, g1i = ta.x * yi + ta.y * yr;
  const float g3r = tb.x * zr - tb.y * zi, g3i = tb.x * zi + tb.y * zr;
  const float d1r = ta.x * (yr - yi), d1i = ta.x * (yr + yi);
  const float d3r = -(ta.x * (zr + zi)), d3i = ta.x * (zr - zi);
  o[1] = diag ? make_float2(d1r, d1i) : make_float2(g1r, g1i);
  o[3] = diag ? make_float2(d3r, d3i) : make_float2(g3r, g3i);
}

// Passes 1-3 of cftfsub/cftbsub for one stream on its 32 lanes.  In: tile holds the 128 complex
// inputs in natural order.  Out: o[t] = element (lam & 15) + 16 t + 64 (lam >> 4).
__device__ __forceinline__ void cft128_passes2(float2* tile, const float* tw2s, int lam,
                                               float2 o[4]) {
  float2 c[4];
  {
    const int rb = (int)(__brev((unsigned)lam) >> 27);
    c[0] = tile[rb];
    c[1] = tile[rb + 64];
    c[2] = tile[rb + 32];
    c[3] = tile[rb + 96];
    bfly4(c, tw2s + (0 * 32 + lam) * 8, o);  // outputs at 4 lam + t
  }
  lds_sync();
  *reinterpret_cast<float4*>(&tile[4 * lam]) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
  *reinterpret_cast<float4*>(&tile[4 * lam + 2]) = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
  lds_sync();
  {
    const int base = 16 * (lam >> 2) + (lam & 3);
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = tile[base + 4 * t];
    bfly4(c, tw2s + (1 * 32 + lam) * 8, o);  // outputs at base + 4 t
    lds_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) tile[base + 4 * t] = o[t];
  }
  lds_sync();
  {
    const int base = 64 * (lam >> 4) + (lam & 15);
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = tile[base + 16 * t];
    bfly4(c, tw2s + (2 * 32 + lam) * 8, o);  // outputs at base + 16 t: the final ownership
  }
}

// radix-2 tail (fft4g.c:939-947 / 989-997): element p (lanes < 16) with p + 64 (lanes >= 16)
__device__ __forceinline__ void radix2_tail(float2 o[4], bool hi, bool backward) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float pr = swz_xor16(o[t].x), pi = swz_xor16(o[t].y);
    // lo: own + partner; hi: partner - own
    const float xr = hi ? pr : o[t].x, xi = hi ? pi : o[t].y;
    const float yr = hi ? -o[t].x : pr, yi = hi ? -o[t].y : pi;
    const float fr = xr + yr, fi = xi + yi;
    o[t] = make_float2(fr, backward ? -fi : fi);
  }
}

// rftfsub / rftbsub (fft4g.c:1234-1283) plus the a[0]/a[1] fix-ups of rdft (fft4g.c:347-352):
// element E = q + 16 t + 64 g pairs with 128 - E; lanes < 16 hold the j side, lanes >= 16 the k side.
__device__ __forceinline__ void real_split2(float2* tile, const float* spls, int lam,
                                            float2 e[4], bool backward) {
  const bool hi = lam >= 16;
  const int base = 64 * (lam >> 4) + (lam & 15);
  lds_sync();
#pragma unroll
  for (int t = 0; t < 4; ++t) tile[base + 16 * t] = e[t];
  lds_sync();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int E = base + 16 * t;
    const float2 pe = tile[(128 - E) & 127];
    const float2 w = *reinterpret_cast<const float2*>(spls + (lam * 4 + t) * 2);  // (wkr, wki)
    const float2 J = hi ? pe : e[t], K = hi ? e[t] : pe;
    const float xr = J.x - K.x, xi = J.y + K.y;
    float2 r;
    if (!backward) {
      const float yr = w.x * xr - w.y * xi, yi = w.x * xi + w.y * xr;
      r = hi ? make_float2(e[t].x + yr, e[t].y - yi) : make_float2(e[t].x - yr, e[t].y - yi);
      if (E == 0) r = make_float2(e[t].x + e[t].y, e[t].x - e[t].y);
      if (E == 64) r = e[t];
    } else {
      const float yr = w.x * xr + w.y * xi, yi = w.x * xi - w.y * xr;
      r = hi ? make_float2(e[t].x + yr, yi - e[t].y) : make_float2(e[t].x - yr, yi - e[t].y);
      if (E == 0) {
        const float h = 0.5f * (e[t].x - e[t].y);
        r = make_float2(e[t].x - h, -h);
      }
      if (E == 64) r = make_float2(e[t].x, -e[t].y);
    }
    e[t] = r;
  }
}

template <bool IO16>
__device__ __forceinline__ void store2(float* y, int idx, float a, float b) {
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

__device__ __forceinline__ float sat16f(float x) {
  return x > 32767 ? 32767 : (x < -32768 ? -32768 : x);
}

// --------------------------------------------------------------------------
// The hand-off build (FLOW), as in ns_kernels1.hip: one launch carries M consecutive frame steps of the whole
// batch (blockIdx.y = step), workgroups are dispatched in grid order, and a per-stream step counter in memory
// orders step k + 1 of a stream behind its own step k.  Every state access is sc1 through ONE buffer resource over
// the batch's state array (the two halves of a wave work on two streams, so the stream's offset is part of the
// lane's offset); frames in / out and the tables stay plain.
struct NsFlowArgs2 {
  unsigned* seq;      // [num_streams]: hand-off steps stream s has completed
  unsigned* abort_w;  // != 0: a wait timed out (1 + stream)
  unsigned want;      // blockIdx.y == 0 processes step `want` of every stream
  int slot0;          // ring slot of that step; step j of the launch uses slot (slot0 + j) % ring
  int ring;
  unsigned per;       // floats between two ring slots of `in` / `out`
};
typedef __attribute__((address_space(1))) unsigned gu32;
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 16;

// The lane's stream: `uni` is a wave-uniform dword offset inside the stream's block, `vec` the lane's.
template <bool FLOW>
struct StateAcc2 {
  float* st;                  // plain build: the stream's block
  __amdgpu_buffer_rsrc_t rs;  // hand-off build: the whole state array
  unsigned sb;                //                 byte offset of the lane's stream in it
  __device__ __forceinline__ StateAcc2(float* state, int stream, int num_streams) {
    st = state + (size_t)stream * aspns::kStreamDwords;
    if constexpr (FLOW) {
      rs = __builtin_amdgcn_make_buffer_rsrc(state, 0, num_streams * (aspns::kStreamDwords * 4), 0x00020000);
      sb = (unsigned)stream * (unsigned)(aspns::kStreamDwords * 4);
    }
  }
  __device__ __forceinline__ float ld1(int uni, int vec) const {
    if constexpr (FLOW) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, sb + vec * 4, uni * 4, kSc1));
    else return st[uni + vec];
  }
  __device__ __forceinline__ float2 ld2(int uni, int vec) const {
    if constexpr (FLOW) {
      const f32x2v v = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(rs, sb + vec * 4, uni * 4, kSc1));
      return make_float2(v.x, v.y);
    } else {
      return *reinterpret_cast<const float2*>(st + uni + vec);
    }
  }
  __device__ __forceinline__ float4 ld4(int uni, int vec) const {
    if constexpr (FLOW) {
      const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, sb + vec * 4, uni * 4, kSc1));
      return make_float4(v.x, v.y, v.z, v.w);
    } else {
      return *reinterpret_cast<const float4*>(st + uni + vec);
    }
  }
  __device__ __forceinline__ void st1(int uni, int vec, float v) const {
    if constexpr (FLOW) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, sb + vec * 4, uni * 4, kSc1);
    else st[uni + vec] = v;
  }
  __device__ __forceinline__ void st2(int uni, int vec, float a, float b) const {
    if constexpr (FLOW) {
      const f32x2v v = {a, b};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), rs, sb + vec * 4, uni * 4, kSc1);
    } else {
      *reinterpret_cast<float2*>(st + uni + vec) = make_float2(a, b);
    }
  }
  __device__ __forceinline__ void st4(int uni, int vec, float4 x) const {
    if constexpr (FLOW) {
      const f32x4v v = {x.x, x.y, x.z, x.w};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), rs, sb + vec * 4, uni * 4, kSc1);
    } else {
      *reinterpret_cast<float4*>(st + uni + vec) = x;
    }
  }
};

// Wait until both streams of the wave (lanes 0-31: `stream` of the low half, 32-63: of the high half; a half past the
// batch's last stream has `mine` false) have completed `want` hand-off steps.  False: given up (abort word set).
__device__ __forceinline__ bool flow_wait2(const NsFlowArgs2& fa, unsigned want, int stream, bool mine, int lane) {
  const gu32* f = (const gu32*)(fa.seq + stream);
  unsigned spins = 0;
  for (;;) {
    const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all(v == want || !mine)) break;
    ++spins;
    if ((spins & 63u) == 0u) {
      const unsigned a = __hip_atomic_load((const gu32*)fa.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_readfirstlane((int)a) != 0) return false;
    }
    if (spins > (1u << 17)) {
      if (lane == 0) __hip_atomic_store((gu32*)fa.abort_w, 1u + (unsigned)stream, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the state loads below the poll
  return true;
}

// launch bounds (256, 3): at most 168 VGPRs, three waves (six streams) per SIMD
template <bool IO16, bool FLOW>
__global__ __launch_bounds__(256, 3) void ns_frame2_kernel(float* __restrict__ state,
                                                        int32_t* __restrict__ hist_all,
                                                        const NsTables* __restrict__ T,
                                                        const float* __restrict__ in,
                                                        float* __restrict__ out,
                                                        int num_streams,
                                                        unsigned long long* __restrict__ stamps, NsFlowArgs2 fa) {
  // diagnostic phase stamps (never passed by the product entry points): wave 0, lane 0
#ifndef NS_STAMP_BLOCK
#define NS_STAMP_BLOCK 0
#endif
#define NS_STAMP(k)                                                                \
  if (stamps != nullptr && threadIdx.x == 0 &&                                     \
      (FLOW ? (blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2) : blockIdx.x == NS_STAMP_BLOCK)) { \
    __builtin_amdgcn_sched_barrier(0);                                             \
    stamps[k] = __builtin_amdgcn_s_memtime();                                      \
    __builtin_amdgcn_sched_barrier(0);                                             \
  }
  NS_STAMP(0)
  __shared__ float2 lds[4][2][128];
  // FFT twiddles (3 passes x 32 lanes x 8) and real-split factors (32 x 4 x 2) staged in LDS once
  // per workgroup: the passes would otherwise stall on a table load from L2 each
  __shared__ __align__(16) float tabs[3 * 32 * 8 + 32 * 4 * 2];
  __shared__ __align__(16) double exp2s[64];  // 2^(j/64) of the lean exp / tanh
  __shared__ __align__(16) double2 logts[128];  // {1/c, log c} of the table-driven log
  __shared__ __align__(16) float wins[kAnal];    // the window, for the synthesis side (ns_core.c:1344-1349)
  // Prologue: every load of the step's first phase is issued before the first wait -- the table
  // pieces first (loads return in order, so the LDS staging waits for them only), then the stream's
  // scalars and samples, which stay in flight across the staging barrier.  Waves past the last stream load from the last pair's addresses and exit
  // after the barrier.
  const float4 tab_v = reinterpret_cast<const float4*>(&T->tw2[0][0][0])[threadIdx.x];  // 256 x 16 B
  const double exp2_v = T->exp2_64[threadIdx.x & 63];
  const double2 logt_v = reinterpret_cast<const double2*>(T->logtab)[threadIdx.x & 127];
  const float4 win_v = reinterpret_cast<const float4*>(T->window)[threadIdx.x & 63];
  const int lane = threadIdx.x & 63;
  const int lam = lane & 31, hb = lane & 32;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.x * 4 + wv;
  const bool pair_live = 2 * pair < num_streams;
  const int stream_raw = 2 * pair + (lane >> 5);
  const int stream = stream_raw < num_streams ? stream_raw : num_streams - 1;  // clamped for the loads
  const bool mine = stream_raw < num_streams;  // (an odd batch leaves the last wave's high half without a stream)
  const StateAcc2<FLOW> sa(state, stream, num_streams);
  int32_t* __restrict__ hist = hist_all + (size_t)stream * kHistDwords;
  unsigned flow_want = 0;
  if constexpr (FLOW) {  // this workgroup's step of the launch: its ring slot, its step number
    const unsigned j = blockIdx.y;
    const unsigned slot = ((unsigned)fa.slot0 + j) % (unsigned)fa.ring;
    in += (size_t)slot * fa.per;
    out += (size_t)slot * fa.per;
    flow_want = fa.want + j;
  }
  bool flow_ok = true;
  float2* tile = lds[wv][lane >> 5];
  const int g = lam >> 4, q = lam & 15;
  const int bin0 = 64 * g + q;  // bin of slot k < 4: bin0 + 16 k

  // ---- scalars: lane L of a half holds scalars L and 32 + L of its stream
  float sv0, sv1;
#define SCF(k) ((k) < 32 ? half_lane(sv0, hb, (k)) : half_lane(sv1, hb, (k)-32))
#define SCI(k) __float_as_int(SCF(k))
#define SET_F(k, val)                                  \
  {                                                    \
    if ((k) < 32) sv0 = (lam == (k)) ? (val) : sv0;    \
    else sv1 = (lam == (k)-32) ? (val) : sv1;          \
  }
#define SET_I(k, val) SET_F(k, __int_as_float(val))

  // ---- sliding analysis buffer [96 carried | 160 new]: lane L owns samples 8L .. 8L+7
  float s8[8];
  if constexpr (FLOW) {
    // the frame's new samples do not depend on the hand-off: requested before the poll; the state follows it
    float i8[8];
    {
      const int li = lam < 12 ? 12 : lam;
      if (!IO16) {
        const float* src = in + (size_t)stream * kBlockL + (8 * li - 96);
        const float4 a = *reinterpret_cast<const float4*>(src);
        const float4 b = *reinterpret_cast<const float4*>(src + 4);
        i8[0] = a.x; i8[1] = a.y; i8[2] = a.z; i8[3] = a.w;
        i8[4] = b.x; i8[5] = b.y; i8[6] = b.z; i8[7] = b.w;
      } else {
        const short* in16 = reinterpret_cast<const short*>(in) + (size_t)stream * kBlockL + (8 * li - 96);
        const short4 a = *reinterpret_cast<const short4*>(in16);
        const short4 b = *reinterpret_cast<const short4*>(in16 + 4);
        i8[0] = (float)a.x; i8[1] = (float)a.y; i8[2] = (float)a.z; i8[3] = (float)a.w;
        i8[4] = (float)b.x; i8[5] = (float)b.y; i8[6] = (float)b.z; i8[7] = (float)b.w;
      }
    }
    if (pair_live) flow_ok = flow_wait2(fa, flow_want, stream, mine, lane);
    sv0 = sa.ld1(kOffScalars, lam);
    sv1 = sa.ld1(kOffScalars + 32, lam);
    const int lh = lam < 12 ? lam : 11;
    const float4 ha = sa.ld4(kOffAnaHist, 8 * lh);
    const float4 hb4 = sa.ld4(kOffAnaHist + 4, 8 * lh);
    const bool hsel = lam < 12;
    s8[0] = hsel ? ha.x : i8[0]; s8[1] = hsel ? ha.y : i8[1];
    s8[2] = hsel ? ha.z : i8[2]; s8[3] = hsel ? ha.w : i8[3];
    s8[4] = hsel ? hb4.x : i8[4]; s8[5] = hsel ? hb4.y : i8[5];
    s8[6] = hsel ? hb4.z : i8[6]; s8[7] = hsel ? hb4.w : i8[7];
  } else {
  float* st = sa.st;
  sv0 = st[kOffScalars + lam];
  sv1 = st[kOffScalars + 32 + lam];
  float* hbuf = st + kOffAnaHist;
  if (!IO16) {
    const float* src = lam < 12 ? hbuf + 8 * lam : in + (size_t)stream * kBlockL + (8 * lam - 96);
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    s8[0] = a.x; s8[1] = a.y; s8[2] = a.z; s8[3] = a.w;
    s8[4] = b.x; s8[5] = b.y; s8[6] = b.z; s8[7] = b.w;
  } else {
    // both sources are read by every lane (clamped), the lane keeps its own: no branch, no wait
    const int lh = lam < 12 ? lam : 11, li = lam < 12 ? 12 : lam;
    const float4 ha = *reinterpret_cast<const float4*>(hbuf + 8 * lh);
    const float4 hb4 = *reinterpret_cast<const float4*>(hbuf + 8 * lh + 4);
    const short* in16 = reinterpret_cast<const short*>(in) + (size_t)stream * kBlockL + (8 * li - 96);
    const short4 a = *reinterpret_cast<const short4*>(in16);
    const short4 b = *reinterpret_cast<const short4*>(in16 + 4);
    const bool hsel = lam < 12;
    s8[0] = hsel ? ha.x : (float)a.x; s8[1] = hsel ? ha.y : (float)a.y;
    s8[2] = hsel ? ha.z : (float)a.z; s8[3] = hsel ? ha.w : (float)a.w;
    s8[4] = hsel ? hb4.x : (float)b.x; s8[5] = hsel ? hb4.y : (float)b.y;
    s8[6] = hsel ? hb4.z : (float)b.z; s8[7] = hsel ? hb4.w : (float)b.w;
  }
  }
  // (pinning the sample loads ahead of the state rows below with a scheduling barrier measured 5 % slower)

#define LOAD5(dst, f)                                                        \
  {                                                                          \
    const float4 v4_ = sa.ld4(kOffVec + (f)*kVecStride, 4 * lam);                        \
    dst[0] = v4_.x; dst[2] = v4_.y; dst[1] = v4_.z; dst[3] = v4_.w; /* row order t = 0, 2, 1, 3 */ \
    dst[4] = SCF(S_TAIL0 + (f));                                             \
  }
// bin 128 of a row lives in scalar slot S_TAIL0 + f (ns_layout.h): read by a broadcast from the
// scalar registers, written back into them
#define STORE5(f, srcv)                                                      \
  if (live) {                                                                \
    sa.st4(kOffVec + (f)*kVecStride, 4 * lam, make_float4(srcv[0], srcv[2], srcv[1], srcv[3])); \
    SET_F(S_TAIL0 + (f), srcv[4])                                            \
  }

  float LQ[3][NS5], DEN[3][NS5], quant[NS5];  // requested once the samples are in, below
  // syntBuf[0..95]: lanes < 16 own samples 2q + 32 t (t = 0..2); every lane loads (no branch, no
  // wait here), the overlap-add uses the owners' values only
  float2 carry[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) carry[t] = sa.ld2(kOffSynt + 32 * t, 2 * q);

  // ---- table staging (the loads above are in flight behind it)
  reinterpret_cast<float4*>(tabs)[threadIdx.x] = tab_v;
  if (threadIdx.x < 64) exp2s[threadIdx.x] = exp2_v;
  if (threadIdx.x < 128) logts[threadIdx.x] = logt_v;
  if (threadIdx.x >= 192) reinterpret_cast<float4*>(wins)[threadIdx.x - 192] = win_v;
  __syncthreads();
  if (!pair_live || !flow_ok) return;
  const float* tw2s = tabs;
  const float* spls = tabs + 3 * 32 * 8;

  // the analysis window comes from the staged copy as well (it used to be 8 KB of L2 reads per
  // workgroup inside the start-of-kernel burst)
  const float4 wa = *reinterpret_cast<const float4*>(wins + 8 * lam);
  const float4 wb = *reinterpret_cast<const float4*>(wins + 8 * lam + 4);
  float wx[8];
  wx[0] = wa.x * s8[0]; wx[1] = wa.y * s8[1]; wx[2] = wa.z * s8[2]; wx[3] = wa.w * s8[3];
  wx[4] = wb.x * s8[4]; wx[5] = wb.y * s8[5]; wx[6] = wb.z * s8[6]; wx[7] = wb.w * s8[7];

  // Windowing + Energy (ns_core.c:969-978, 951-960)
  float epart = wx[0] * wx[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) epart += wx[k] * wx[k];
  const float energy1 = half_sum(epart);
  const bool live = energy1 != 0.0f && mine;  // ns_core.c:1072-1082 / 1239-1264 (a half without a stream stores nothing)

  // the carried 96 samples of the next frame are this frame's last 96
  if (lam >= 20 && mine) {
    sa.st4(kOffAnaHist, 8 * lam - 160, make_float4(s8[0], s8[1], s8[2], s8[3]));  // (the uniform part of an offset is never negative)
    sa.st4(kOffAnaHist + 4, 8 * lam - 160, make_float4(s8[4], s8[5], s8[6], s8[7]));
  }

  // the tracker rows are requested once the frame's samples are in (a shorter start-of-kernel burst:
  // every wave of the launch starts at once, and under that load a request takes 1.5 us and more);
  // they are used after the transform, the magnitudes and the logarithms
  LOAD5(LQ[0], V_LQ0) LOAD5(LQ[1], V_LQ1) LOAD5(LQ[2], V_LQ2)
  LOAD5(DEN[0], V_DEN0) LOAD5(DEN[1], V_DEN1) LOAD5(DEN[2], V_DEN2)
  LOAD5(quant, V_QUANT)
  NS_STAMP(1)
  // ---- forward FFT (ns_core.c:886-911)
  *reinterpret_cast<float4*>(&tile[4 * lam]) = make_float4(wx[0], wx[1], wx[2], wx[3]);
  *reinterpret_cast<float4*>(&tile[4 * lam + 2]) = make_float4(wx[4], wx[5], wx[6], wx[7]);
  lds_sync();
  float2 el[4];
  cft128_passes2(tile, tw2s, lam, el);
  radix2_tail(el, g == 1, false);
  real_split2(tile, spls, lam, el, false);

  NS_STAMP(2)
  // second group of state rows (latency hides under magnitude / log / trackers; requesting them
  // in the prologue as well made the start-of-kernel burst longer and the step 0.6 us slower)
  float smooth[NS5], noisePrev[NS5], magnPrevA[NS5], logLrt[NS5], avgPause[NS5];
  LOAD5(magnPrevA, V_MAGNPREV_A) LOAD5(logLrt, V_LOGLRT) LOAD5(avgPause, V_AVGPAUSE)
  LOAD5(smooth, V_SMOOTH) LOAD5(noisePrev, V_NOISEPREV)

  float re[NS5], im[NS5], magn[NS5];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    re[k] = el[k].x;
    im[k] = el[k].y;
  }
  re[4] = half_lane(el[0].y, hb, 0);  // R128 sits in the imaginary slot of element 0
  im[4] = 0.f;
  if (lam == 0) im[0] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) magn[k] = re[k] * re[k] + im[k] * im[k];
  {
    float m2[4] = {magn[0], magn[1], magn[2], magn[3]}, rt[4];
    fsqrt_n<4>(m2, rt);
#pragma unroll
    for (int k = 0; k < 4; ++k) magn[k] = rt[k] + 1.f;
  }
  if (lam == 0) magn[0] = fabsf(re[0]) + 1.f;
  magn[4] = fabsf(re[4]) + 1.f;

  // per-lane partial of a per-bin quantity: 4 owned bins in lane order, then the tail on lane 0
  // sum over the 129 bins of a per-bin quantity: the lane's four owned bins in lane order, the 32-lane
  // butterfly over the half-wave's partials, then bin 128 (association ASP_NS_REDUCE_TREE32 of oracle/ns_oracle.c:
  // bin 128 joins the butterfly's result, as in the pair-layout kernel -- joined to lane 0's partial first, one of
  // the eight golden streams left the 1e-4 bar of the reference's own left-to-right sum)
#define SUM5(v) (half_sum(((v[0] + v[1]) + v[2]) + v[3]) + v[4])

  int blockInd = SCI(S_BLOCKIND);
  const float overdrive = SCF(S_OVERDRIVE);
  const float denoiseBound = SCF(S_DENOISEBOUND);
  float priorSpeechProb = SCF(S_PRIORSPEECHPROB);
  const int gainmap = SCI(S_GAINMAP);

  float noise[NS5], prevStsa[NS5];
  blockInd++;  // ns_core.c:1084 (committed only for live streams)
  const int updateParsFlag = SCI(S_MUP0);
  int updates = SCI(S_UPDATES);
  int counter[3] = {SCI(S_COUNTER0), SCI(S_COUNTER1), SCI(S_COUNTER2)};

  float lmagn[NS5];
  log_f32_via_tab_n<NS5>(magn, lmagn, logts);

  NS_STAMP(3)
  float signalEnergy, sumMagn;
  {
    float se[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) se[k] = re[k] * re[k] + im[k] * im[k];
    signalEnergy = SUM5(se);
    sumMagn = SUM5(magn);
    signalEnergy = DIV129(signalEnergy);
  }

  NS_STAMP(4)
  // ---- NoiseEstimation (ns_core.c:217-285)
  if (updates < NS_END_STARTUP_LONG) updates++;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float cnt = (float)counter[s];
    const float cnt1 = (float)(counter[s] + 1);
    const float rcnt1 = 1.f / cnt1;
    {
      // the five bins of the lane as packed pairs (ns_device.h: F5); same operations as per bin
      F5 den(DEN[s]), lq(LQ[s]);
      const F5 lm(lmagn);
      const F5 dq = fdiv5v(F5(NS_FACTOR * 1.f), den);  // used where density > 1
      const F5 delta = sel5(gt5(den, F5(1.0f)), dq, F5(NS_FACTOR));
      const B5 up = gt5(lm, lq);
      const F5 step = div_by_uniform5(sel5(up, NS_QUANTILE * delta, (1.f - NS_QUANTILE) * delta), cnt1, rcnt1);
      lq = sel5(up, lq + step, lq - step);
      const F5 nd = div_by_uniform5(cnt * den + 1.f / (2.f * NS_WIDTH), cnt1, rcnt1);
      den = sel5(lt5(abs5(lm - lq), F5(NS_WIDTH)), nd, den);
      den.store(DEN[s]);
      lq.store(LQ[s]);
    }
    if (counter[s] >= NS_END_STARTUP_LONG) {
      counter[s] = 0;
      if (updates >= NS_END_STARTUP_LONG) {
#pragma unroll
        for (int k = 0; k < NS5; ++k) quant[k] = exp_f32_via_f64(LQ[s][k], exp2s);
      }
    }
    counter[s]++;
  }
  if (updates < NS_END_STARTUP_LONG) {
#pragma unroll
    for (int k = 0; k < NS5; ++k) quant[k] = exp_f32_via_f64(LQ[2][k], exp2s);
  }
#pragma unroll
  for (int k = 0; k < NS5; ++k) noise[k] = quant[k];
  STORE5(V_LQ0, LQ[0]) STORE5(V_LQ1, LQ[1]) STORE5(V_LQ2, LQ[2])
  STORE5(V_DEN0, DEN[0]) STORE5(V_DEN1, DEN[1]) STORE5(V_DEN2, DEN[2])
  STORE5(V_QUANT, quant)

  NS_STAMP(5)
  // ---- startup noise model (ns_core.c:1091-1100, 1109-1162)
  float whiteNoiseLevel = SCF(S_WHITE);
  float pinkNoiseNumerator = SCF(S_PINKNUM);
  float pinkNoiseExp = SCF(S_PINKEXP);
  float fd5 = SCF(S_FD5);
  const bool startup = blockInd < NS_END_STARTUP_SHORT;
  if (startup) {
    float lm5[NS5], lilm[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      const int bin = k < 4 ? bin0 + 16 * k : 128;
      const float li = T->logi[bin];
      lm5[k] = bin >= NS_START_BAND ? lmagn[k] : 0.f;
      lilm[k] = bin >= NS_START_BAND ? li * lmagn[k] : 0.f;
    }
    const float sum_log_magn = SUM5(lm5);
    const float sum_log_i_log_magn = SUM5(lilm);
    const float sum_log_i = T->sum_log_i, sum_log_i_square = T->sum_log_i_square;
    whiteNoiseLevel += DIV129(sumMagn) * overdrive;
    float tmpFloat1 = sum_log_i_square * ((float)(kBins - NS_START_BAND));
    tmpFloat1 -= (sum_log_i * sum_log_i);
    float tmpFloat2 = (sum_log_i_square * sum_log_magn - sum_log_i * sum_log_i_log_magn);
    float tmpFloat3 = tmpFloat2 / tmpFloat1;
    if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
    pinkNoiseNumerator += tmpFloat3;
    tmpFloat2 = (sum_log_i * sum_log_magn);
    tmpFloat2 -= ((float)(kBins - NS_START_BAND)) * sum_log_i_log_magn;
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
    float pn[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      const int bin = k < 4 ? bin0 + 16 * k : 128;
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
    STORE5(V_PARAMNOISE, pn)
  }
  if (blockInd < NS_END_STARTUP_LONG) {  // ns_core.c:1165-1169
    fd5 *= blockInd;
    fd5 += signalEnergy;
    fd5 /= (blockInd + 1);
  }

  NS_STAMP(6)
  // ---- ComputeSnr (ns_core.c:566-588)
  float snrLocPost[NS5], snrLocPrior[NS5];
  {
    float dn1[NS5], dn2[NS5], q1[NS5], q2[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      dn1[k] = noisePrev[k] + 0.0001f;
      dn2[k] = noise[k] + 0.0001f;
    }
    fdiv5(magnPrevA, dn1, q1);
    fdiv5(magn, dn2, q2);  // used where magn > noise
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      const float previousEstimateStsa = q1[k] * smooth[k];
      prevStsa[k] = previousEstimateStsa;
      snrLocPost[k] = 0.f;
      if (magn[k] > noise[k]) snrLocPost[k] = q2[k] - 1.f;
      snrLocPrior[k] = NS_DD_PR_SNR * previousEstimateStsa + (1.f - NS_DD_PR_SNR) * snrLocPost[k];
    }
  }

  NS_STAMP(7)
  // ---- ComputeSpectralFlatness (ns_core.c:523-556)
  float fd0 = SCF(S_FD0), fd4 = SCF(S_FD4), fd6 = SCF(S_FD6);
  {
    float fl5[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) fl5[k] = (k == 0 && lam == 0) ? 0.f : lmagn[k];
    float num = SUM5(fl5);
    float den = sumMagn - half_lane(magn[0], hb, 0);
    den = DIV129(den);
    num = DIV129(num);
    const float spectralTmp = fdiv(exp_f32_via_f64(num, exp2s), den);
    fd0 += NS_SPECT_FL_TAVG * (spectralTmp - fd0);
  }
  // ---- ComputeSpectralDifference (ns_core.c:595-634)
  {
    float avgPauseMean = SUM5(avgPause);
    float avgMagn = sumMagn;
    avgPauseMean = DIV129(avgPauseMean);
    avgMagn = DIV129(avgMagn);
    float cv[NS5], vp[NS5], vm[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      const float dm = magn[k] - avgMagn, dp = avgPause[k] - avgPauseMean;
      cv[k] = dm * dp;
      vp[k] = dp * dp;
      vm[k] = dm * dm;
    }
    float covMagnPause = SUM5(cv);
    float varPause = SUM5(vp);
    float varMagn = SUM5(vm);
    covMagnPause = DIV129(covMagnPause);
    varPause = DIV129(varPause);
    varMagn = DIV129(varMagn);
    fd6 += signalEnergy;
    float avgDiffNormMagn = varMagn - fdiv(covMagnPause * covMagnPause, varPause + 0.0001f);
    avgDiffNormMagn = fdiv(avgDiffNormMagn, fd5 + 0.0001f);
    fd4 += NS_SPECT_DIFF_TAVG * (avgDiffNormMagn - fd4);
  }

  NS_STAMP(8)
  // ---- histograms / prior model (FeatureUpdate, ns_core.c:766-790)
  float fd3 = SCF(S_FD3);
  PriorModel pm;
  pm.p0 = SCF(S_PMP0);
  pm.p1 = SCF(S_PMP1);
  pm.p3 = SCF(S_PMP3);
  pm.p4 = SCF(S_PMP4);
  pm.p5 = SCF(S_PMP5);
  pm.p6 = SCF(S_PMP6);
  const float pmp2 = SCF(S_PMP2);
  int mup0 = updateParsFlag, mup3 = SCI(S_MUP3);
  const int mup1 = SCI(S_MUP1);
  bool window_closed = false;
  if (updateParsFlag >= 1) {
    mup3--;
    {  // FeatureParameterExtraction(self, 0), ns_core.c:309-334: lanes 0..2 of a half take one
       // histogram each (LRT, spectral flatness, spectral difference); one writer per bin and stream,
       // so a no-return atomic add is the increment without the load -> add -> store round trip
      const float fv = lam == 0 ? fd3 : (lam == 1 ? fd0 : fd4);
      const float bw = lam == 1 ? 0.05f : 0.1f, rbw = lam == 1 ? 1.0f / 0.05f : 1.0f / 0.1f;
      const float lim = lam == 1 ? kHist * 0.05f : kHist * 0.1f;
      if (mup3 > 0 && lam < 3 && live && (fv < lim) && (fv >= 0.0f))
        atomicAdd(&hist[lam * kHistStride + (int)div_by_uniform(fv, bw, rbw)], 1);
    }
  }
  {
    // the window close needs the whole wave for one stream: do it stream by stream
    const bool closing = updateParsFlag >= 1 && mup3 == 0 && live;
    const unsigned long long closing_mask = __ballot(closing);  // per-stream flag: bits 0 and 32
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool c_h = ((closing_mask >> (32 * h)) & 1ull) != 0;  // wave-uniform
      if (c_h) {
        PriorModel pin;
        pin.p0 = __shfl(pm.p0, 32 * h, 64);
        pin.p1 = __shfl(pm.p1, 32 * h, 64);
        pin.p3 = __shfl(pm.p3, 32 * h, 64);
        pin.p4 = __shfl(pm.p4, 32 * h, 64);
        pin.p5 = __shfl(pm.p5, 32 * h, 64);
        pin.p6 = __shfl(pm.p6, 32 * h, 64);
        const int w1 = __shfl(mup1, 32 * h, 64), f0 = __shfl(mup0, 32 * h, 64);
        int32_t* hh = hist_all + (size_t)(2 * pair + h) * kHistDwords;
        const PriorModel po = close_histogram_window<FLOW>(hh, lane, w1, f0 >= 1, pin);
        if ((lane >> 5) == h) pm = po;
      }
    }
    if (updateParsFlag >= 1 && mup3 == 0) {
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

  NS_STAMP(9)
  // ---- SpeechNoiseProb (ns_core.c:642-749)
  {
    float t1[NS5], lt1[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) t1[k] = 1.f + 2.f * snrLocPrior[k];
    log_f32_via_tab_n<NS5>(t1, lt1, logts);
    float tn[NS5], td5[NS5], t2v[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      tn[k] = 2.f * snrLocPrior[k];
      td5[k] = t1[k] + 0.0001f;
    }
    fdiv5(tn, td5, t2v);
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      const float t2 = t2v[k];
      const float besselTmp = (snrLocPost[k] + 1.f) * t2;
      logLrt[k] += NS_LRT_TAVG * (besselTmp - lt1[k] - logLrt[k]);
    }
  }
  float logLrtTimeAvgKsum = SUM5(logLrt);
  logLrtTimeAvgKsum = DIV129(logLrtTimeAvgKsum);
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
    // the three tanh() of :696-725 on lanes 0..2 of each half: one call serves both streams
    const float arg = lam == 0 ? arg0 : (lam == 1 ? arg1 : arg2);
    const float th = tanh_f32_via_f64(arg, exp2s);
    const float indicator0 = 0.5f * (half_lane(th, hb, 0) + 1.f);
    const float indicator1 = 0.5f * (half_lane(th, hb, 1) + 1.f);
    const float indicator2 = 0.5f * (half_lane(th, hb, 2) + 1.f);
    const float indPrior = pm.p4 * indicator0 + pm.p5 * indicator1 + pm.p6 * indicator2;
    priorSpeechProb += NS_PRIOR_UPDATE * (indPrior - priorSpeechProb);
    if (priorSpeechProb > 1.f) priorSpeechProb = 1.f;
    if (priorSpeechProb < 0.01f) priorSpeechProb = 0.01f;
  }
  float probSpeech[NS5];
  {
    const float gainPrior = fdiv(1.f - priorSpeechProb, priorSpeechProb + 0.0001f);
    float nl[NS5], ev[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) nl[k] = -logLrt[k];
    exp_f32_via_f64_n<NS5>(nl, ev, exp2s);
    {
      float pd[NS5];
      const float ones[NS5] = {1.f, 1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int k = 0; k < NS5; ++k) {
        float invLrt = ev[k];
        invLrt = (float)gainPrior * invLrt;
        pd[k] = 1.f + invLrt;
      }
      fdiv5(ones, pd, probSpeech);
    }
  }

  NS_STAMP(10)
  // ---- UpdateNoiseEstimate (ns_core.c:800-846): the time constant carried into bin i is the
  // one bin i-1 selected.  Bin i-1 lives on the previous lane of the 16-lane row, same slot;
  // at a row start it is lane 15 / 31 of the half, previous slot (bin 63 for bin 64).
  {
    // row_ror:1 hands lane q the value of lane q - 1 of its 16-lane row and lane 0 that of lane 15:
    // slot k of the rotated values is the predecessor for q > 0, slot k - 1 for q == 0
    float prevProb[NS5], ror[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ror[k] = dpp_move<0x121>(probSpeech[k]);  // row_ror:1
    const float l15_3 = half_lane(probSpeech[3], hb, 15), l31_3 = half_lane(probSpeech[3], hb, 31);
    prevProb[0] = q == 0 ? (g == 0 ? 0.0f : l15_3) : ror[0];  // bin 0 has no predecessor; bin 64 <- bin 63
#pragma unroll
    for (int k = 1; k < 4; ++k) prevProb[k] = q == 0 ? ror[k - 1] : ror[k];
    prevProb[4] = l31_3;  // bin 128 <- bin 127
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      float gammaOld = prevProb[k] > NS_PROB_RANGE ? NS_SPEECH_UPDATE : NS_NOISE_UPDATE;
      if (k == 0 && lam == 0) gammaOld = NS_NOISE_UPDATE;
      const float ps = probSpeech[k], pns = 1.f - probSpeech[k];
      const float noiseUpdateTmp =
          gammaOld * noisePrev[k] + (1.f - gammaOld) * (pns * magn[k] + ps * noisePrev[k]);
      float gammaNew = NS_NOISE_UPDATE;
      if (ps > NS_PROB_RANGE) gammaNew = NS_SPEECH_UPDATE;
      if (ps < NS_PROB_RANGE) avgPause[k] += NS_GAMMA_PAUSE * (magn[k] - avgPause[k]);
      float nz;
      if (gammaNew == gammaOld) {
        nz = noiseUpdateTmp;
      } else {
        nz = gammaNew * noisePrev[k] + (1.f - gammaNew) * (pns * magn[k] + ps * noisePrev[k]);
        if (noiseUpdateTmp < nz) nz = noiseUpdateTmp;
      }
      noise[k] = nz;
    }
  }
  STORE5(V_LOGLRT, logLrt) STORE5(V_AVGPAUSE, avgPause)
  STORE5(V_MAGNPREV_A, magn)  // ns_core.c:1180 (== magnPrevProcess while paired)

  NS_STAMP(11)
  // ---- Process: decision-directed Wiener gain (ns_core.c:985-1007, 1276-1307)
  float initMagn[NS5], pnoise[NS5];
  if (startup) {  // ns_core.c:1268-1272
    LOAD5(initMagn, V_INITMAGN)
    LOAD5(pnoise, V_PARAMNOISE)
#pragma unroll
    for (int k = 0; k < NS5; ++k) initMagn[k] += magn[k];
    STORE5(V_INITMAGN, initMagn)
  }
  float gainv[NS5];
  float gq1[NS5], gq2[NS5], snrP[NS5];
  {
    float gd1[NS5], gd2[NS5];
#pragma unroll
    for (int k = 0; k < NS5; ++k) gd1[k] = noise[k] + 0.0001f;
    fdiv5(magn, gd1, gq1);  // used where magn > noise
#pragma unroll
    for (int k = 0; k < NS5; ++k) {
      float currentEstimateStsa = 0.f;
      if (magn[k] > noise[k]) currentEstimateStsa = gq1[k] - 1.f;
      snrP[k] = NS_DD_PR_SNR * prevStsa[k] + (1.f - NS_DD_PR_SNR) * currentEstimateStsa;
      gd2[k] = overdrive + snrP[k];
    }
    fdiv5(snrP, gd2, gq2);
  }
#pragma unroll
  for (int k = 0; k < NS5; ++k) {
    float gg = gq2[k];
    if (gg < denoiseBound) gg = denoiseBound;
    if (gg > 1.f) gg = 1.f;
    if (startup) {
      float tmp = (initMagn[k] - overdrive * pnoise[k]);
      tmp /= (initMagn[k] + 0.0001f);
      if (tmp < denoiseBound) tmp = denoiseBound;
      if (tmp > 1.f) tmp = 1.f;
      gg *= (blockInd);
      tmp *= (NS_END_STARTUP_SHORT - blockInd);
      gg += tmp;
      gg /= (NS_END_STARTUP_SHORT);
    }
    gainv[k] = gg;
    re[k] *= gg;
    im[k] *= gg;
  }
  STORE5(V_SMOOTH, gainv)      // ns_core.c:1304
  STORE5(V_NOISEPREV, noise)   // ns_core.c:1310

  NS_STAMP(12)
  // ---- IFFT (ns_core.c:923-944)
#pragma unroll
  for (int k = 0; k < 4; ++k) el[k] = make_float2(re[k], im[k]);
  if (lam == 0) el[0].y = re[4];  // Ooura packing: a[1] = R128
  real_split2(tile, spls, lam, el, true);
  lds_sync();
  {
    const int base = 64 * g + q;
#pragma unroll
    for (int t = 0; t < 4; ++t) tile[base + 16 * t] = el[t];
  }
  lds_sync();
  cft128_passes2(tile, tw2s, lam, el);
  radix2_tail(el, g == 1, true);
  float td[8];  // samples 2E, 2E+1 of elements E = q + 16 t + 64 g
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    td[2 * t] = el[t].x * (2.f / kAnal);
    td[2 * t + 1] = el[t].y * (2.f / kAnal);
  }

  NS_STAMP(13)
  // ---- energy-based gain compensation (ns_core.c:1315-1342)
  float factor = 1.f;
  if (gainmap == 1 && blockInd > NS_END_STARTUP_LONG) {
    float factor1 = 1.f, factor2 = 1.f;
    float e2 = td[0] * td[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) e2 += td[k] * td[k];
    const float energy2 = half_sum(e2);
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
  {
    float* y = IO16 ? reinterpret_cast<float*>(reinterpret_cast<short*>(out) + (size_t)stream * kBlockL)
                    : out + (size_t)stream * kBlockL;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = 2 * q + 32 * t + 128 * g;  // sample index of td[2t]
      const float2 w = *reinterpret_cast<const float2*>(wins + n);
      const float c0 = (g == 0 && t < 3) ? carry[t < 3 ? t : 0].x : 0.f;
      const float c1 = (g == 0 && t < 3) ? carry[t < 3 ? t : 0].y : 0.f;
      float o0 = c0 + factor * (w.x * td[2 * t]);
      float o1 = c1 + factor * (w.y * td[2 * t + 1]);
      if (!live) {  // zero input: emit the synthesis tail, clear it (ns_core.c:1239-1264)
        o0 = c0;
        o1 = c1;
        if (n >= 160) {
          o0 = 0.f;
          o1 = 0.f;
        }
      }
      if (n >= 160) {
        if (mine) sa.st2(kOffSynt, n - 160, o0, o1);
      } else {
        if (mine) store2<IO16>(y, n, sat16f(o0), sat16f(o1));
      }
    }
  }

  NS_STAMP(14)
  // ---- commit scalars (only live streams advance)
  if (live) {
    SET_I(S_UPDATES, updates);
    SET_I(S_COUNTER0, counter[0]);
    SET_I(S_COUNTER1, counter[1]);
    SET_I(S_COUNTER2, counter[2]);
    SET_I(S_MUP0, mup0);
    SET_I(S_MUP3, mup3);
    SET_F(S_SIGNALENERGY, signalEnergy);
    SET_F(S_SUMMAGN, sumMagn);
    if (startup) {
      SET_F(S_WHITE, whiteNoiseLevel);
      SET_F(S_PINKNUM, pinkNoiseNumerator);
      SET_F(S_PINKEXP, pinkNoiseExp);
    }
    if (window_closed) {
      SET_F(S_PMP0, pm.p0);
      SET_F(S_PMP1, pm.p1);
      SET_F(S_PMP3, pm.p3);
      SET_F(S_PMP4, pm.p4);
      SET_F(S_PMP5, pm.p5);
      SET_F(S_PMP6, pm.p6);
    }
    SET_F(S_FD0, fd0);
    SET_F(S_FD3, fd3);
    SET_F(S_FD4, fd4);
    SET_F(S_FD5, fd5);
    SET_F(S_FD6, fd6);
    SET_I(S_BLOCKIND, blockInd);
    SET_F(S_PRIORSPEECHPROB, priorSpeechProb);
    if (mine) {
      sa.st1(kOffScalars, lam, sv0);
      sa.st1(kOffScalars + 32, lam, sv1);
    }
  }
  NS_STAMP(15)
  if constexpr (FLOW) {  // publish both streams' step: every store of this wave drained first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lam == 0 && mine)
      __hip_atomic_store((gu32*)(fa.seq + stream), flow_want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#undef NS_STAMP
#undef SCF
#undef SCI
#undef SET_F
#undef SET_I
#undef LOAD5
#undef STORE5
#undef SUM5
}

}  // namespace

namespace aspns {

hipError_t launch_ns_frame2(bool io16, float* state, int32_t* hist, const NsTables* T,
                            const float* in, float* out, int num_streams, hipStream_t s,
                            unsigned long long* stamps) {
  // two streams per wave, eight per 256-thread workgroup; an odd count leaves the last wave's high half idle
  const dim3 grid(((num_streams + 1) / 2 + 3) / 4), block(256);
  const NsFlowArgs2 none = {nullptr, nullptr, 0u, 0, 1, 0u};
  if (io16)
    hipLaunchKernelGGL((ns_frame2_kernel<true, false>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, none);
  else
    hipLaunchKernelGGL((ns_frame2_kernel<false, false>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, none);
  return hipGetLastError();
}

// `steps` consecutive frame steps of the hand-off build in one launch (grid y = step), as launch_ns_frame1_flow
hipError_t launch_ns_frame2_flow(bool io16, float* state, int32_t* hist, const NsTables* T,
                                 const float* in, float* out, int num_streams, hipStream_t s,
                                 unsigned* seq, unsigned* abort_w, unsigned want, int steps, int slot0, int ring,
                                 size_t per, unsigned long long* stamps) {
  const int gx = (((num_streams + 1) / 2 + 3) / 4 + 7) / 8 * 8;  // a multiple of 8: see NsFlowArgs of ns_kernels1.hip
  const dim3 grid(gx, steps), block(256);
  const NsFlowArgs2 fa = {seq, abort_w, want, slot0, ring, (unsigned)per};
  if (io16)
    hipLaunchKernelGGL((ns_frame2_kernel<true, true>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, fa);
  else
    hipLaunchKernelGGL((ns_frame2_kernel<false, true>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, fa);
  return hipGetLastError();
}

}  // namespace aspns
