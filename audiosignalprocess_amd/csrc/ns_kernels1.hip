// ns_kernels1.hip -- the fused Analyze+Process frame step with ONE stream per wave64 and two
// bins per lane: the low-latency build of the frame step.
//
// Same arithmetic as ns_frame_kernel<true,true> (ns_kernels.hip) -- every per-bin float operation of
// ns_core.c:1043-1359 in the reference's order, Ooura-order FFT (fft4g.c), exact libm forms
// (ns_device.h) -- mapped so that a frame step of 4096 streams is 4096 short waves, all resident at
// once (four per SIMD at <= 128 VGPRs): a step is bound by the time a wave needs from its first load
// to its last store plus the launch boundary (profiles/README.md, rounds 2 and 3).
//
//   * lane L = 2 lam + h (lam = q + 16 g: the "dual lane" of ns_layout.h's row order) owns the
//     FFT elements / bins E = q + 64 g + 16 t for t = h and t = h + 2: exactly the two outputs
//     of its half of a radix-4 butterfly (outputs 0, 2 on even lanes, 1, 3 on odd lanes), and,
//     with ns_layout.h's row order (t = 0, 2, 1, 3 inside a dual lane), one 8-byte access per
//     state row; bin 128 is computed on every lane (wave-uniform) and committed with the scalars;
//   * per-stream scalars are wave-uniform: read from the scalar row with v_readlane, every
//     data-independent branch of the reference is a scalar branch, written back with v_writelane;
//   * the three radix-4 passes go through a 1 KB LDS tile per wave (each lane computes half a
//     butterfly, no duplicated arithmetic), the radix-2 tail through v_permlane32_swap (element p
//     on lane L, p + 64 on lane L ^ 32), the real split through one LDS gather;
//   * cross-bin sums: lane-local (slot A + slot B, + bin 128 on lane 0), then the wave64 xor
//     butterfly of ns_device.h's wave_sum; oracle/ns_oracle.c reproduces this association as
//     ASP_NS_REDUCE_TREE64P and the tests compare bit for bit (outputs and every state array).
#include <hip/hip_runtime.h>

#include "ns_device.h"
#include "ns_layout.h"
#include "ns_pair_fft.h"

namespace {
using namespace aspns_dev;
using namespace aspns_pair;

// which CU a wave runs on, for the timeline diagnostic: HW_ID[15:8] (cu_id, sh_id, se_id) and XCC_ID[3:0]
__device__ __forceinline__ unsigned long long ns_cu_tag() {
  const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);    // HW_REG_HW_ID, bits 8..15
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID, bits 0..3
  return (unsigned long long)((xcc << 8) | hw);
}

constexpr int NS3 = 3;  // 2 owned bins + the tail bin 128

// ---- the hand-off build (FLOW): consecutive frame steps overlap on the chip.
// One launch carries M consecutive frame steps of the whole batch: blockIdx.y is the step, blockIdx.x the
// group of four streams.  Workgroups are dispatched in linear order (x fastest), so every workgroup of
// step j has been dispatched before the first one of step j + 1: the wave that takes stream s in step
// j + 1 may therefore WAIT for the wave that has stream s in step j -- that wave is resident or done,
// whatever else runs on the chip (the argument of a decoupled look-back scan).  What orders the two is a
// word in memory: the wave that has finished stream s of step k stores seq[s] = k + 1 after draining its
// stores; the wave that takes stream s in step k + 1 polls seq[s] before its first state load.  Every
// state access of this build is an sc1 access (write-through stores, loads that bypass the CU's L1), the
// form MI355X_MICROARCH.md's visibility section lists for hand-offs without an agent-scope fence per wave;
// `in` / `out` frames and the constant tables are not handed off and stay plain.  The state still goes
// through memory every step (SURVEY 8(d)'s frame-synchronous model: nothing of a stream stays on chip
// between its steps); what disappears is the chip-wide phase lock of one launch per step (all waves load,
// then all compute, then all store) and the idle time at every launch boundary.  The x extent of the grid
// is a multiple of 8, so that (with workgroups dealt round-robin to the 8 XCDs) the two workgroups of a
// stream's consecutive steps come from the same XCD's in-order share of the grid.  The wait is bounded
// all the same: a wave that gives up sets the abort word, which every later wait sees, and the host
// reports the failure (ns_api.hip, flow_check).
struct NsFlowArgs {
  unsigned* seq;      // [num_streams]: number of hand-off steps stream s has completed
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
constexpr int kSc1 = 16;  // cache-policy operand of the buffer intrinsics: sc1

// One stream's state block: `uni` is a wave-uniform dword offset, `vec` the lane's dword offset.
template <bool FLOW>
struct StateAcc {
  float* st;
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ explicit StateAcc(float* p) : st(p) {
    if constexpr (FLOW) rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, aspns::kStreamDwords * 4, 0x00020000);
  }
  __device__ __forceinline__ float ld1(int uni, int vec) const {
    if constexpr (FLOW) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vec * 4, uni * 4, kSc1));
    else return st[uni + vec];
  }
  __device__ __forceinline__ float2 ld2(int uni, int vec) const {
    if constexpr (FLOW) {
      const f32x2v v = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(rs, vec * 4, uni * 4, kSc1));
      return make_float2(v.x, v.y);
    } else {
      return *reinterpret_cast<const float2*>(st + uni + vec);
    }
  }
  __device__ __forceinline__ float4 ld4(int uni, int vec) const {
    if constexpr (FLOW) {
      const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, vec * 4, uni * 4, kSc1));
      return make_float4(v.x, v.y, v.z, v.w);
    } else {
      return *reinterpret_cast<const float4*>(st + uni + vec);
    }
  }
  __device__ __forceinline__ void st1(int uni, int vec, float v) const {
    if constexpr (FLOW) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vec * 4, uni * 4, kSc1);
    else st[uni + vec] = v;
  }
  __device__ __forceinline__ void st2(int uni, int vec, float a, float b) const {
    if constexpr (FLOW) {
      const f32x2v v = {a, b};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), rs, vec * 4, uni * 4, kSc1);
    } else {
      *reinterpret_cast<float2*>(st + uni + vec) = make_float2(a, b);
    }
  }
  __device__ __forceinline__ void st4(int uni, int vec, float4 x) const {
    if constexpr (FLOW) {
      const f32x4v v = {x.x, x.y, x.z, x.w};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), rs, vec * 4, uni * 4, kSc1);
    } else {
      *reinterpret_cast<float4*>(st + uni + vec) = x;
    }
  }
};

// Wait until stream `stream` has completed `want` hand-off steps.  Returns false when the wait was given
// up (the abort word is set: by this wave after ~0.1 s of polling, or by another one before).
__device__ __forceinline__ bool flow_wait(const NsFlowArgs& fa, unsigned want, int stream, int lane) {
  const gu32* f = (const gu32*)(fa.seq + stream);
  unsigned spins = 0;
  for (;;) {
    const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) == want) break;
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
  // no instruction: keeps the compiler from moving the state loads above the poll
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return true;
}


template <bool IO16, bool FLOW>
__global__ __launch_bounds__(256, 4) void ns_frame1_kernel(float* __restrict__ state,
                                                           int32_t* __restrict__ hist_all,
                                                           const NsTables* __restrict__ T,
                                                           const float* __restrict__ in,
                                                           float* __restrict__ out,
                                                           int num_streams,
                                                           unsigned long long* __restrict__ stamps,
                                                           int stamp_mode, NsFlowArgs fa) {
#ifdef NS1_BUDGET
  // instruction-budget build (tools/ns_valu_budget.py, never shipped): the phase marks become
  // assembly comments and the steady-state conditions are asserted, so that the straight-line
  // code between two marks is what a wave executes per frame after start-up
#define NS_STAMP(k)                                                                \
  __builtin_amdgcn_sched_barrier(0);                                               \
  asm volatile("; NS_PHASE " #k);                                                  \
  __builtin_amdgcn_sched_barrier(0);
#define NS_STEADY(x) __builtin_assume(x)
#else
  // diagnostic stamps (never passed by the product entry points).  stamp_mode 0: the 16 phase stamps
  // (shader clock) of workgroup 0's first wave (hand-off build: of the workgroup in the middle of the
  // launch, x = grid / 2, y = steps / 2, plus a 17th once its stores have drained).  stamp_mode 1 ("timeline"): every workgroup's first
  // wave records the 100 MHz real-time counter at its start, after its first loads, before its last
  // stores and at its end (4 values per workgroup) -- the launch-level picture.
#define NS_STAMP(k)                                                                          \
  if (stamps != nullptr && threadIdx.x == 0) {                                               \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (stamp_mode != 0) {                                                                   \
      if ((k) == 0 || (k) == 1 || (k) == 14 || (k) == 15)                                    \
        stamps[blockIdx.x * 4 + ((k) == 0 ? 0 : (k) == 1 ? 1 : (k) == 14 ? 2 : 3)] =        \
            __builtin_amdgcn_s_memrealtime() | ((k) == 0 ? ns_cu_tag() << 48 : 0ull);       \
    } else if (FLOW ? (blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2)           \
                    : blockIdx.x == 0) {                                                     \
      stamps[k] = __builtin_amdgcn_s_memtime();                                              \
    }                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  }
#define NS_STEADY(x)
#endif
  NS_STAMP(0)
  __shared__ __align__(16) float2 lds[4][130];  // 128 elements + the slot lane 0 reads past them (ns_pair_fft.h)
  // per-lane twiddles of the three passes (3 x 64 x 4), real-split factors (32 x 4 x 2) and the
  // window, staged in LDS once per workgroup behind the state loads
  __shared__ __align__(16) float tabs[3 * 64 * 4 + 32 * 4 * 2];
  __shared__ __align__(16) float wins[kAnal];
  __shared__ __align__(16) double exp2s[64];     // 2^(j/64) of the lean exp / tanh
  __shared__ __align__(16) double2 logts[128];   // {1/c, log c} of the table-driven log
  const int tid = threadIdx.x;
  // ---- prologue: every load of the first phase is issued before the first wait (table pieces
  // first: loads return in order, the LDS staging waits for them only)
  const float4* tab_src = tid < 192 ? reinterpret_cast<const float4*>(&T->tw[0][0][0]) + tid
                                    : reinterpret_cast<const float4*>(&T->spl[0][0][0]) + (tid - 192);
  const float4 tab_v = *tab_src;
  const float4 win_v = reinterpret_cast<const float4*>(T->window)[tid & 63];
  const double exp2_v = T->exp2_64[tid & 63];
  const double2 logt_v = reinterpret_cast<const double2*>(T->logtab)[tid & 127];
  const int lane = tid & 63;
  const int diagbits = T->diag[lane];
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* tile = lds[wv];
  const int lam = lane >> 1, h = lane & 1;
  const int g = lam >> 4, q = lam & 15;
  const int binA = q + 64 * g + 16 * h;  // slot 0; slot 1 = binA + 32; slot 2 = bin 128
  const uint32_t gmask = g ? 0x80000000u : 0u;
  const int stream_raw = blockIdx.x * 4 + wv;
  bool wave_live = stream_raw < num_streams;
  const int stream = wave_live ? stream_raw : num_streams - 1;  // clamped for the loads
  float* __restrict__ st = state + (size_t)stream * kStreamDwords;
  int32_t* __restrict__ hist = hist_all + (size_t)stream * kHistDwords;
  const StateAcc<FLOW> sa(st);
  unsigned flow_want = 0;
  if constexpr (FLOW) {  // this workgroup's step of the launch: its ring slot, its step number
    const unsigned j = blockIdx.y;
    const unsigned slot = ((unsigned)fa.slot0 + j) % (unsigned)fa.ring;
    in += (size_t)slot * fa.per;
    out += (size_t)slot * fa.per;
    flow_want = fa.want + j;
  }

  // ---- scalars: lane k holds scalar k (wave-uniform values, read with v_readlane)
  float sv;
#define SC_I(k) __builtin_amdgcn_readlane(__float_as_int(sv), (k))
#define SC_F(k) __int_as_float(SC_I(k))
#define SC_SET_I(k, val) sv = writelane_bits<(k)>(sv, (int)(val))
#define SC_SET_F(k, val) sv = setlane_vgpr<(k)>(sv, (val))
  // ---- sliding analysis buffer [96 carried | 160 new]: lane L owns samples 4L .. 4L+3
  float4 s4;
  // syntBuf[0..95]: the lane's output samples 2E, 2E+1 that still carry overlap are those of
  // slot 0 when g == 0 (2q + 32h) and of slot 1 when g == 0 and h == 0 (2q + 64); every lane
  // loads (no branch), the overlap-add uses the owners' values only
  float2 carryA, carryB;
  if constexpr (FLOW) {
    // the frame's new samples do not depend on the hand-off: requested before the poll; the state follows it
    float4 s4in = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!IO16) {
      if (lane >= 24) s4in = *reinterpret_cast<const float4*>(in + (size_t)stream * kBlockL + 4 * (lane - 24));
    } else {
      const int li = lane < 24 ? 24 : lane;
      const short* in16 = reinterpret_cast<const short*>(in) + (size_t)stream * kBlockL + 4 * (li - 24);
      const short4 a = *reinterpret_cast<const short4*>(in16);
      s4in = make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w);
    }
    if (wave_live) wave_live = flow_wait(fa, flow_want, stream, lane);
    sv = sa.ld1(kOffScalars, lane);
    const float4 ha = sa.ld4(kOffAnaHist, 4 * (lane < 24 ? lane : 23));
    const bool hsel = lane < 24;
    s4.x = hsel ? ha.x : s4in.x;
    s4.y = hsel ? ha.y : s4in.y;
    s4.z = hsel ? ha.z : s4in.z;
    s4.w = hsel ? ha.w : s4in.w;
    carryA = sa.ld2(kOffSynt, 2 * q + 32 * h);
    carryB = sa.ld2(kOffSynt, 2 * q + 64);
  } else {
    sv = st[kOffScalars + lane];
    float* hbuf = st + kOffAnaHist;
    if (!IO16) {
      const float* src = lane < 24 ? hbuf + 4 * lane : in + (size_t)stream * kBlockL + 4 * (lane - 24);
      s4 = *reinterpret_cast<const float4*>(src);
    } else {
      const int lh = lane < 24 ? lane : 23, li = lane < 24 ? 24 : lane;
      const float4 ha = *reinterpret_cast<const float4*>(hbuf + 4 * lh);
      const short* in16 = reinterpret_cast<const short*>(in) + (size_t)stream * kBlockL + 4 * (li - 24);
      const short4 a = *reinterpret_cast<const short4*>(in16);
      const bool hsel = lane < 24;
      s4.x = hsel ? ha.x : (float)a.x;
      s4.y = hsel ? ha.y : (float)a.y;
      s4.z = hsel ? ha.z : (float)a.z;
      s4.w = hsel ? ha.w : (float)a.w;
    }
    carryA = *reinterpret_cast<const float2*>(st + kOffSynt + 2 * q + 32 * h);
    carryB = *reinterpret_cast<const float2*>(st + kOffSynt + 2 * q + 64);
  }

  // ---- table staging (the loads above are in flight behind it)
  reinterpret_cast<float4*>(tabs)[tid] = tab_v;
  if (tid < 64) exp2s[tid] = exp2_v;
  if (tid < 128) logts[tid] = logt_v;
  if (tid >= 192) reinterpret_cast<float4*>(wins)[tid - 192] = win_v;
  __syncthreads();
  if (!wave_live) return;
  const float* tws = tabs;
  const float* spls = tabs + 3 * 64 * 4;
  const PairFftLane fl = pair_fft_lane(lane, diagbits);

#define LOADV(dst, f)                                                                          \
  {                                                                                            \
    const float2 v2_ = sa.ld2(kOffVec + (f)*kVecStride, 2 * lane);                             \
    dst[0] = v2_.x; dst[1] = v2_.y;                                                            \
  }
#define LOADT(dst, f) dst[2] = SC_F(S_TAIL0 + (f));
#define LOAD3(dst, f) LOADV(dst, f) LOADT(dst, f)
#define STORE3(f, srcv)                                                                        \
  {                                                                                            \
    sa.st2(kOffVec + (f)*kVecStride, 2 * lane, srcv[0], srcv[1]);                              \
    SC_SET_F(S_TAIL0 + (f), srcv[2]);                                                          \
  }
  // the step is done for this stream: the hand-off build publishes it, every store of this wave drained first
#define NS_STREAM_DONE()                                                                       \
  if constexpr (FLOW) {                                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
    NS_STAMP(16)                                                                               \
    if (lane == 0)                                                                             \
      __hip_atomic_store((gu32*)(fa.seq + stream), flow_want + 1u, __ATOMIC_RELAXED,           \
                         __HIP_MEMORY_SCOPE_AGENT);                                            \
  }                                                                                            \
  return;

  // state rows are requested in two groups, just ahead of their use (requesting all of them before
  // the first wait measured slower: every wave of a launch starts at once, and a bigger
  // start-of-kernel burst makes every wave wait longer)
  float LQ[3][NS3], DEN[3][NS3], quant[NS3];
  float smooth[NS3], noisePrev[NS3], magnPrevA[NS3], logLrt[NS3], avgPause[NS3];


  const float4 w4 = *reinterpret_cast<const float4*>(wins + 4 * lane);
  const float wx0 = w4.x * s4.x, wx1 = w4.y * s4.y, wx2 = w4.z * s4.z, wx3 = w4.w * s4.w;
  // Windowing + Energy (ns_core.c:969-978, 951-960)
  float epart = wx0 * wx0;
  epart += wx1 * wx1;
  epart += wx2 * wx2;
  epart += wx3 * wx3;
  const float energy1 = wave_sum_bcast(epart);

  // the carried 96 samples of the next frame are this frame's last 96
  if (lane >= 40) sa.st4(kOffAnaHist, 4 * (lane - 40), s4);

  NS_STEADY(energy1 != 0.0f);
  if (energy1 == 0.0f) {
    // Analyze: nothing but the buffer slide (ns_core.c:1072-1082); Process: emit the synthesis
    // tail and clear it (ns_core.c:1239-1264)
    float* y = IO16 ? reinterpret_cast<float*>(reinterpret_cast<short*>(out) + (size_t)stream * kBlockL)
                    : out + (size_t)stream * kBlockL;
    float2 o01 = make_float2(0.f, 0.f);
    if (lane < 48) o01 = sa.ld2(kOffSynt, 2 * lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    store2p<IO16>(y, 2 * lane, sat16p(o01.x), sat16p(o01.y));
    if (lane < 16) store2p<IO16>(y, 128 + 2 * lane, 0.f, 0.f);
    if (lane < 48) sa.st2(kOffSynt, 2 * lane, 0.f, 0.f);
    NS_STREAM_DONE()
  }

  // the tracker rows are requested once the frame's samples are in; they are used after the
  // transform, the magnitudes and the logarithms
  LOADV(LQ[0], V_LQ0) LOADV(LQ[1], V_LQ1) LOADV(LQ[2], V_LQ2)
  LOADV(DEN[0], V_DEN0) LOADV(DEN[1], V_DEN1) LOADV(DEN[2], V_DEN2)
  LOADV(quant, V_QUANT)
  LOADT(LQ[0], V_LQ0) LOADT(LQ[1], V_LQ1) LOADT(LQ[2], V_LQ2)
  LOADT(DEN[0], V_DEN0) LOADT(DEN[1], V_DEN1) LOADT(DEN[2], V_DEN2)
  LOADT(quant, V_QUANT)
  NS_STAMP(1)
  // ---- forward FFT (ns_core.c:886-911)
  *reinterpret_cast<float4*>(&tile[2 * lane]) = make_float4(wx0, wx1, wx2, wx3);
  lds_sync1();
  f32x2 er, ei;  // the lane's two bins {slot 0, slot 1}: real parts, imaginary parts
  {
    f32x2 ea, eb;
    cft128_passes1(tile, tws, fl, lane, ea, eb);
    radix2_tail1(ea, eb, gmask, false, er, ei);
  }
  real_split1(tile, spls, lane, er, ei, false);

  NS_STAMP(2)
  // second group of state rows (latency hides under magnitude / log / trackers)
  LOADV(magnPrevA, V_MAGNPREV_A) LOADV(logLrt, V_LOGLRT) LOADV(avgPause, V_AVGPAUSE)
  LOADV(smooth, V_SMOOTH) LOADV(noisePrev, V_NOISEPREV)
  LOADT(magnPrevA, V_MAGNPREV_A) LOADT(logLrt, V_LOGLRT) LOADT(avgPause, V_AVGPAUSE)
  LOADT(smooth, V_SMOOTH) LOADT(noisePrev, V_NOISEPREV)

  float re[NS3], im[NS3], magn[NS3];
  re[0] = er.x;
  im[0] = ei.x;
  re[1] = er.y;
  im[1] = ei.y;
  re[2] = lane_bcast(ei.x, 0);  // R128 sits in the imaginary slot of element 0 (lane 0, slot 0)
  im[2] = 0.f;
  if (lane == 0) im[0] = 0.f;
  {
    float m2[2] = {re[0] * re[0] + im[0] * im[0], re[1] * re[1] + im[1] * im[1]}, rt[2];
    fsqrt_n<2>(m2, rt);
    magn[0] = rt[0] + 1.f;
    magn[1] = rt[1] + 1.f;
  }
  if (lane == 0) magn[0] = fabsf(re[0]) + 1.f;
  magn[2] = fabsf(re[2]) + 1.f;

  // sum over the 129 bins of a per-bin quantity: the lane's two owned bins, the wave64 butterfly over the
  // 64 partials, then bin 128 (association ASP_NS_REDUCE_TREE64P of oracle/ns_oracle.c)
#define SUM3(v) (wave_sum_bcast(v[0] + v[1]) + v[2])

  int blockInd = SC_I(S_BLOCKIND);
  const float overdrive = SC_F(S_OVERDRIVE);
  const float denoiseBound = SC_F(S_DENOISEBOUND);
  float priorSpeechProb = SC_F(S_PRIORSPEECHPROB);
  const int gainmap = SC_I(S_GAINMAP);

  float noise[NS3], prevStsa[NS3];
  blockInd++;  // ns_core.c:1084
  const int updateParsFlag = SC_I(S_MUP0);
  int updates = SC_I(S_UPDATES);
  int counter[3] = {SC_I(S_COUNTER0), SC_I(S_COUNTER1), SC_I(S_COUNTER2)};
  // steady state (budget build only): past both start-up windows, no tracker publishes, the
  // histogram window stays open, gain compensation on
  NS_STEADY(blockInd > NS_END_STARTUP_LONG + 1);
  NS_STEADY(updates >= NS_END_STARTUP_LONG);
  NS_STEADY(counter[0] < NS_END_STARTUP_LONG - 1 && counter[1] < NS_END_STARTUP_LONG - 1 && counter[2] < NS_END_STARTUP_LONG - 1);
  NS_STEADY(counter[0] >= 0 && counter[1] >= 0 && counter[2] >= 0);
  NS_STEADY(updateParsFlag >= 1);
  NS_STEADY(gainmap == 1);

  float lmagn[NS3];
  log_f32_via_tab_n<NS3>(magn, lmagn, logts);

  NS_STAMP(3)
  // the four cross-bin sums that need only this frame's spectrum and the loaded rows, reduced side by
  // side: signal energy (ns_core.c:1089-1103), sum of magnitudes, the flatness numerator (bins 1..128,
  // :535-541) and the mean of magnAvgPause (:603-607)
  float signalEnergy, sumMagn, flatNum, avgPauseMean;
  {
    float p_se = (re[0] * re[0] + im[0] * im[0]) + (re[1] * re[1] + im[1] * im[1]);
    float p_sm = magn[0] + magn[1];
    float p_fl = lane == 0 ? lmagn[1] : lmagn[0] + lmagn[1];
    float p_ap = avgPause[0] + avgPause[1];
    wave_sums_bcast(p_se, p_sm, p_fl, p_ap);
    signalEnergy = p_se + (re[2] * re[2] + im[2] * im[2]);
    sumMagn = p_sm + magn[2];
    flatNum = p_fl + lmagn[2];
    avgPauseMean = p_ap + avgPause[2];
    signalEnergy = DIV129(signalEnergy);
  }

  NS_STAMP(4)
  // ---- NoiseEstimation (ns_core.c:217-285)
  if (updates < NS_END_STARTUP_LONG) updates++;
  bool quant_new = false;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float cnt = (float)counter[s];
    const float cnt1 = (float)(counter[s] + 1);
    const float rcnt1 = fdiv(1.f, cnt1);  // == 1.f / cnt1 (cnt1 = 1 .. 201)
    {
      // ns_core.c:232-260 with two of its three branches folded into the arithmetic (same roundings):
      // delta = FACTOR / max(density, 1) (the quotient by 1 is exact), and the step carries its sign,
      // lq += (+QUANTILE delta) / n or (-(1 - QUANTILE) delta) / n (products, quotients and x + (-y)
      // are sign-symmetric)
      F3 den(DEN[s]), lq(LQ[s]);
      const F3 lm(lmagn);
      const F3 delta = fdiv3v(F3(NS_FACTOR * 1.f), max3(den, 1.0f));
      const F3 coef = sel3(gt3(lm, lq), F3(NS_QUANTILE), F3(-(1.f - NS_QUANTILE)));
      lq = lq + div_by_uniform3(coef * delta, cnt1, rcnt1);
      const F3 nd = div_by_uniform3(cnt * den + 1.f / (2.f * NS_WIDTH), cnt1, rcnt1);
      den = sel3(lt3(abs3(lm - lq), F3(NS_WIDTH)), nd, den);
      den.store(DEN[s]);
      lq.store(LQ[s]);
    }
    if (counter[s] >= NS_END_STARTUP_LONG) {
      counter[s] = 0;
      if (updates >= NS_END_STARTUP_LONG) {
        exp_f32_via_f64_n<NS3>(LQ[s], quant, exp2s);
        quant_new = true;
      }
    }
    counter[s]++;
  }
  if (updates < NS_END_STARTUP_LONG) {
    exp_f32_via_f64_n<NS3>(LQ[2], quant, exp2s);
    quant_new = true;
  }
#pragma unroll
  for (int k = 0; k < NS3; ++k) noise[k] = quant[k];
  STORE3(V_LQ0, LQ[0]) STORE3(V_LQ1, LQ[1]) STORE3(V_LQ2, LQ[2])
  STORE3(V_DEN0, DEN[0]) STORE3(V_DEN1, DEN[1]) STORE3(V_DEN2, DEN[2])
  // the published quantile changes once in ~67 frames past start-up (a tracker publishes every 200
  // frames, ns_core.c:262-270): its row is written back only then (wave-uniform branch)
  if (quant_new) STORE3(V_QUANT, quant)

  NS_STAMP(5)
  // ---- startup noise model (ns_core.c:1091-1100, 1109-1162)
  float whiteNoiseLevel = SC_F(S_WHITE);
  float pinkNoiseNumerator = SC_F(S_PINKNUM);
  float pinkNoiseExp = SC_F(S_PINKEXP);
  float fd5 = SC_F(S_FD5);
  const bool startup = blockInd < NS_END_STARTUP_SHORT;
  if (startup) {
    float lm3[NS3], lilm[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      const int bin = k < 2 ? binA + 32 * k : 128;
      const float li = T->logi[bin];
      lm3[k] = bin >= NS_START_BAND ? lmagn[k] : 0.f;
      lilm[k] = bin >= NS_START_BAND ? li * lmagn[k] : 0.f;
    }
    const float sum_log_magn = SUM3(lm3);
    const float sum_log_i_log_magn = SUM3(lilm);
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
    float pn[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      const int bin = k < 2 ? binA + 32 * k : 128;
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
    STORE3(V_PARAMNOISE, pn)
  }
  if (blockInd < NS_END_STARTUP_LONG) {  // ns_core.c:1165-1169
    fd5 *= blockInd;
    fd5 += signalEnergy;
    fd5 /= (blockInd + 1);
  }

  NS_STAMP(6)
  // ---- ComputeSnr (ns_core.c:566-588)
  float snrLocPost[NS3], snrLocPrior[NS3];
  {
    float dn1[NS3], dn2[NS3], q1[NS3], q2[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      dn1[k] = noisePrev[k] + 0.0001f;
      dn2[k] = noise[k] + 0.0001f;
    }
    fdiv3(magnPrevA, dn1, q1);
    fdiv3(magn, dn2, q2);  // used where magn > noise
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      const float previousEstimateStsa = q1[k] * smooth[k];
      prevStsa[k] = previousEstimateStsa;
      snrLocPost[k] = 0.f;
      if (magn[k] > noise[k]) snrLocPost[k] = q2[k] - 1.f;
      snrLocPrior[k] = NS_DD_PR_SNR * previousEstimateStsa + (1.f - NS_DD_PR_SNR) * snrLocPost[k];
    }
  }

  NS_STAMP(7)
  // ---- ComputeSpectralFlatness (ns_core.c:523-556)
  float fd0 = SC_F(S_FD0), fd4 = SC_F(S_FD4), fd6 = SC_F(S_FD6);
  {
    float num = flatNum;
    float den = sumMagn - lane_bcast(magn[0], 0);
    den = DIV129(den);
    num = DIV129(num);
    const float spectralTmp = fdiv(exp_f32_via_f64(num, exp2s), den);
    fd0 += NS_SPECT_FL_TAVG * (spectralTmp - fd0);
  }
  // ---- ComputeSpectralDifference (ns_core.c:595-634)
  {
    float avgMagn = sumMagn;
    avgPauseMean = DIV129(avgPauseMean);
    avgMagn = DIV129(avgMagn);
    float cv[NS3], vp[NS3], vm[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      const float dm = magn[k] - avgMagn, dp = avgPause[k] - avgPauseMean;
      cv[k] = dm * dp;
      vp[k] = dp * dp;
      vm[k] = dm * dm;
    }
    float covMagnPause = cv[0] + cv[1], varPause = vp[0] + vp[1], varMagn = vm[0] + vm[1];
    wave_sums_bcast(covMagnPause, varPause, varMagn);
    covMagnPause += cv[2];
    varPause += vp[2];
    varMagn += vm[2];
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
  const int mup1 = SC_I(S_MUP1);
  NS_STEADY(mup3 > 2);
  bool window_closed = false;
  if (updateParsFlag >= 1) {
    mup3--;
    if (mup3 > 0) {
      // FeatureParameterExtraction(self, 0), ns_core.c:309-334: lanes 0..2 take one histogram each
      // (LRT, spectral flatness, spectral difference); one writer per bin and stream, so a
      // no-return atomic add is the increment without the load -> add -> store round trip
      const float fv = lane == 0 ? fd3 : (lane == 1 ? fd0 : fd4);
      const float bw = lane == 1 ? 0.05f : 0.1f, rbw = lane == 1 ? 1.0f / 0.05f : 1.0f / 0.1f;
      const float lim = lane == 1 ? kHist * 0.05f : kHist * 0.1f;
      if (lane < 3 && (fv < lim) && (fv >= 0.0f))
        atomicAdd(&hist[lane * kHistStride + (int)div_by_uniform(fv, bw, rbw)], 1);  // agent scope (sc1)
    }
    if (mup3 == 0) {
      pm = close_histogram_window<FLOW>(hist, lane, mup1, mup0 >= 1, pm);
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
    float t1[NS3], lt1[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) t1[k] = 1.f + 2.f * snrLocPrior[k];
    log_f32_via_tab_n<NS3>(t1, lt1, logts);
    float tn[NS3], td3[NS3], t2v[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      tn[k] = 2.f * snrLocPrior[k];
      td3[k] = t1[k] + 0.0001f;
    }
    fdiv3(tn, td3, t2v);
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      const float t2 = t2v[k];
      const float besselTmp = (snrLocPost[k] + 1.f) * t2;
      logLrt[k] += NS_LRT_TAVG * (besselTmp - lt1[k] - logLrt[k]);
    }
  }
  float logLrtTimeAvgKsum = SUM3(logLrt);
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
    // the three tanh() of :696-725 evaluated on lanes 0..2 of one call
    const float arg = lane == 0 ? arg0 : (lane == 1 ? arg1 : arg2);
    const float th = tanh_f32_via_f64(arg, exp2s);
    const float indicator0 = 0.5f * (lane_bcast(th, 0) + 1.f);
    const float indicator1 = 0.5f * (lane_bcast(th, 1) + 1.f);
    const float indicator2 = 0.5f * (lane_bcast(th, 2) + 1.f);
    const float indPrior = pm.p4 * indicator0 + pm.p5 * indicator1 + pm.p6 * indicator2;
    priorSpeechProb += NS_PRIOR_UPDATE * (indPrior - priorSpeechProb);
    if (priorSpeechProb > 1.f) priorSpeechProb = 1.f;
    if (priorSpeechProb < 0.01f) priorSpeechProb = 0.01f;
  }
  float probSpeech[NS3];
  {
    const float gainPrior = fdiv(1.f - priorSpeechProb, priorSpeechProb + 0.0001f);
    float nl[NS3], ev[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) nl[k] = -logLrt[k];
    exp_f32_via_f64_n<NS3>(nl, ev, exp2s);
    {
      float pd[NS3];
      const float ones[NS3] = {1.f, 1.f, 1.f};
#pragma unroll
      for (int k = 0; k < NS3; ++k) {
        float invLrt = ev[k];
        invLrt = (float)gainPrior * invLrt;
        pd[k] = 1.f + invLrt;
      }
      fdiv3(ones, pd, probSpeech);
    }
  }

  NS_STAMP(10)
  // ---- UpdateNoiseEstimate (ns_core.c:800-846): the time constant carried into bin i is the one
  // bin i-1 selected.  For q > 0 bin i-1 is the same slot of lane L - 2; for q == 0 it is bin
  // 15 + 16 (t - 1) + 64 g (t > 0) or bin 63 (bin 64), i.e. a slot of lane 30 / 31 (+ 32 g):
  //   slot 0 (t = h):     h = 1: lane 30 + 32 g slot 0;   h = 0, g = 1: lane 31 slot 1;   bin 0: none
  //   slot 1 (t = h + 2): h = 0: lane 31 + 32 g slot 0;   h = 1: lane 30 + 32 g slot 1
  {
    const int srcA = q > 0 ? lane - 2 : (h ? 30 + 32 * g : 31);
    const int srcB = q > 0 ? lane - 2 : (h ? 30 + 32 * g : 31 + 32 * g);
    const bool a_from1 = q == 0 && h == 0;  // slot 0 takes the source lane's slot 1
    const bool b_from1 = q > 0 || h == 1;   // slot 1 takes the source lane's slot 1
    const float a0 = __shfl(probSpeech[0], srcA, 64), a1 = __shfl(probSpeech[1], srcA, 64);
    const float b0 = __shfl(probSpeech[0], srcB, 64), b1 = __shfl(probSpeech[1], srcB, 64);
    float prevProb[NS3];
    prevProb[0] = a_from1 ? a1 : a0;
    prevProb[1] = b_from1 ? b1 : b0;
    prevProb[2] = lane_bcast(probSpeech[1], 63);  // bin 128 <- bin 127 (q = 15, g = 1, t = 3)
    // ns_core.c:813-845.  The update with a time constant g is u(g) = g noisePrev + (1 - g) x, x = (1 -
    // ps) magn + ps noisePrev; the reference computes u(gammaOld) and, when gammaNew differs, keeps the
    // smaller of u(gammaOld) and u(gammaNew).  With both constants' updates at hand that is: the old
    // bin's choice, the new bin's choice, their minimum (equal choices give the same value twice).
    {
      const F3 np(noisePrev), mg(magn), ps(probSpeech), ap(avgPause);
      const F3 x = (1.f - ps) * mg + ps * np;
      const F3 uS = NS_SPEECH_UPDATE * np + (1.f - NS_SPEECH_UPDATE) * x;
      const F3 uN = NS_NOISE_UPDATE * np + (1.f - NS_NOISE_UPDATE) * x;
      B3 oldSpeech = gt3(F3(prevProb), F3(NS_PROB_RANGE));
      oldSpeech.v[0] = oldSpeech.v[0] && lane != 0;  // bin 0 has no predecessor: gamma = NOISE_UPDATE
      const F3 uOld = sel3(oldSpeech, uS, uN);
      const F3 uNew = sel3(gt3(ps, F3(NS_PROB_RANGE)), uS, uN);
      min3(uOld, uNew).store(noise);
      sel3(lt3(ps, F3(NS_PROB_RANGE)), ap + NS_GAMMA_PAUSE * (mg - ap), ap).store(avgPause);
    }
  }
  STORE3(V_LOGLRT, logLrt) STORE3(V_AVGPAUSE, avgPause)
  STORE3(V_MAGNPREV_A, magn)  // ns_core.c:1180 (== magnPrevProcess while paired)

  NS_STAMP(11)
  // ---- Process: decision-directed Wiener gain (ns_core.c:985-1007, 1276-1307)
  float initMagn[NS3], pnoise[NS3];
  if (startup) {  // ns_core.c:1268-1272
    LOAD3(initMagn, V_INITMAGN)
    LOAD3(pnoise, V_PARAMNOISE)
#pragma unroll
    for (int k = 0; k < NS3; ++k) initMagn[k] += magn[k];
    STORE3(V_INITMAGN, initMagn)
  }
  float gainv[NS3];
  float gq1[NS3], gq2[NS3], snrP[NS3];
  {
    float gd1[NS3], gd2[NS3];
#pragma unroll
    for (int k = 0; k < NS3; ++k) gd1[k] = noise[k] + 0.0001f;
    fdiv3(magn, gd1, gq1);  // used where magn > noise
#pragma unroll
    for (int k = 0; k < NS3; ++k) {
      float currentEstimateStsa = 0.f;
      if (magn[k] > noise[k]) currentEstimateStsa = gq1[k] - 1.f;
      snrP[k] = NS_DD_PR_SNR * prevStsa[k] + (1.f - NS_DD_PR_SNR) * currentEstimateStsa;
      gd2[k] = overdrive + snrP[k];
    }
    fdiv3(snrP, gd2, gq2);
  }
#pragma unroll
  for (int k = 0; k < NS3; ++k) {
    float gg = fmin_raw(fmax_raw(gq2[k], denoiseBound), 1.f);  // ns_core.c:1001-1006
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
  STORE3(V_SMOOTH, gainv)      // ns_core.c:1304
  STORE3(V_NOISEPREV, noise)   // ns_core.c:1310

  NS_STAMP(12)
  // ---- IFFT (ns_core.c:923-944)
  er = f32x2{re[0], re[1]};
  ei = f32x2{im[0], im[1]};
  if (lane == 0) ei.x = re[2];  // Ooura packing: a[1] = R128
  real_split1(tile, spls, lane, er, ei, true);
  lds_sync1();
  {
    const int base = 64 * g + q + 16 * h;
    float* tf = reinterpret_cast<float*>(tile);
    tf[2 * base] = er.x;
    tf[2 * base + 1] = ei.x;
    tf[2 * base + 64] = er.y;
    tf[2 * base + 65] = ei.y;
  }
  lds_sync1();
  f32x2 tr, ti;  // samples 2E (tr) and 2E + 1 (ti) of elements E = binA (slot 0) and binA + 32 (slot 1)
  {
    f32x2 ea, eb;
    cft128_passes1(tile, tws, fl, lane, ea, eb);
    radix2_tail1(ea, eb, gmask, true, tr, ti);
  }
  const float td0 = tr.x * (2.f / kAnal), td1 = ti.x * (2.f / kAnal);
  const float td2 = tr.y * (2.f / kAnal), td3s = ti.y * (2.f / kAnal);

  NS_STAMP(13)
  // ---- energy-based gain compensation (ns_core.c:1315-1342)
  float factor = 1.f;
  if (gainmap == 1 && blockInd > NS_END_STARTUP_LONG) {
    float factor1 = 1.f, factor2 = 1.f;
    float e2 = td0 * td0;
    e2 += td1 * td1;
    e2 += td2 * td2;
    e2 += td3s * td3s;
    const float energy2 = wave_sum_bcast(e2);
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
    const int nA = 2 * binA, nB = nA + 64;  // sample index of td0 / td2
    const float2 wA = *reinterpret_cast<const float2*>(wins + nA);
    const float2 wB = *reinterpret_cast<const float2*>(wins + nB);
    const float cA0 = g == 0 ? carryA.x : 0.f, cA1 = g == 0 ? carryA.y : 0.f;
    const float cB0 = (g == 0 && h == 0) ? carryB.x : 0.f, cB1 = (g == 0 && h == 0) ? carryB.y : 0.f;
    const float oA0 = cA0 + factor * (wA.x * td0), oA1 = cA1 + factor * (wA.y * td1);
    const float oB0 = cB0 + factor * (wB.x * td2), oB1 = cB1 + factor * (wB.y * td3s);
    if (nA >= 160) {
      sa.st2(kOffSynt, nA - 160, oA0, oA1);
    } else {
      store2p<IO16>(y, nA, sat16p(oA0), sat16p(oA1));
    }
    if (nB >= 160) {
      sa.st2(kOffSynt, nB - 160, oB0, oB1);
    } else {
      store2p<IO16>(y, nB, sat16p(oB0), sat16p(oB1));
    }
  }

  NS_STAMP(14)
  // ---- commit scalars
  SC_SET_I(S_UPDATES, updates);
  SC_SET_I(S_COUNTER0, counter[0]);
  SC_SET_I(S_COUNTER1, counter[1]);
  SC_SET_I(S_COUNTER2, counter[2]);
  SC_SET_I(S_MUP0, mup0);
  SC_SET_I(S_MUP3, mup3);
  SC_SET_F(S_SIGNALENERGY, signalEnergy);
  SC_SET_F(S_SUMMAGN, sumMagn);
  if (startup) {
    SC_SET_F(S_WHITE, whiteNoiseLevel);
    SC_SET_F(S_PINKNUM, pinkNoiseNumerator);
    SC_SET_F(S_PINKEXP, pinkNoiseExp);
  }
  if (window_closed) {
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
  sa.st1(kOffScalars, lane, sv);
  NS_STAMP(15)
  NS_STREAM_DONE()
#undef NS_STREAM_DONE
#undef NS_STAMP
#undef SC_I
#undef SC_F
#undef SC_SET_I
#undef SC_SET_F
#undef LOAD3
#undef LOADV
#undef LOADT
#undef STORE3
#undef SUM3
}

}  // namespace

namespace aspns {

hipError_t launch_ns_frame1(bool io16, float* state, int32_t* hist, const NsTables* T,
                            const float* in, float* out, int num_streams, hipStream_t s,
                            unsigned long long* stamps, int stamp_mode) {
  const dim3 grid((num_streams + 3) / 4), block(256);
  const NsFlowArgs none = {nullptr, nullptr, 0u, 0, 1, 0u};
  if (io16)
    hipLaunchKernelGGL((ns_frame1_kernel<true, false>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, stamp_mode, none);
  else
    hipLaunchKernelGGL((ns_frame1_kernel<false, false>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, stamp_mode, none);
  return hipGetLastError();
}

// `steps` consecutive frame steps of the hand-off build in one launch: steps want .. want + steps - 1 of every
// stream (seq[s] == want on entry, want + steps on exit); step j reads / writes ring slot (slot0 + j) % ring of
// in / out (slots `per` floats apart).
hipError_t launch_ns_frame1_flow(bool io16, float* state, int32_t* hist, const NsTables* T,
                                 const float* in, float* out, int num_streams, hipStream_t s,
                                 unsigned* seq, unsigned* abort_w, unsigned want, int steps, int slot0, int ring,
                                 size_t per, unsigned long long* stamps) {
  const int gx = ((num_streams + 3) / 4 + 7) / 8 * 8;
  const dim3 grid(gx, steps), block(256);
  const NsFlowArgs fa = {seq, abort_w, want, slot0, ring, (unsigned)per};
  if (io16)
    hipLaunchKernelGGL((ns_frame1_kernel<true, true>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, 0, fa);
  else
    hipLaunchKernelGGL((ns_frame1_kernel<false, true>), grid, block, 0, s, state, hist, T, in, out,
                       num_streams, stamps, 0, fa);
  return hipGetLastError();
}

}  // namespace aspns
