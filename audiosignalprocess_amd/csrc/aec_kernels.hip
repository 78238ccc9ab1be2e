// aec_kernels.hip -- gfx950 kernels of the batched acoustic echo canceller.  Replaces, for many
// independent streams per launch,
//   WebRtcAec_BufferFarendPartition / TimeToFrequency           (aec_core.c:1618-1635, 772-795)
//   ProcessBlock: FilterFar, ScaleErrorSignal, FilterAdaptation (aec_core.c:1084-1287, 147-269)
//   NonLinearProcessing: SubbandCoherence, SmoothedPSD, PartitionDelay, OverdriveAndSuppress,
//   ComfortNoise                                                (aec_core.c:852-1082, 271-500)
//   aec_rdft_forward_128 / aec_rdft_inverse_128                 (aec_rdft.c:124-556)
// and the ring-buffer copies of WebRtcAec_BufferFarend / WebRtcAec_ProcessFrames whose positions
// the host control plane (aec_api.hip) computes.
//
// Mapping: one wave64 per stream, four streams per 256-thread workgroup, no workgroup barrier
// after the table staging.  Per-bin work puts bin q on lane q (bin 64 is a second trip of
// lane 0); the local arrays of the reference live in a 9 KB LDS region private to the wave.
// A 128-point real FFT is 16 radix-4 butterflies per pass, so four transforms run side by side
// in a wave (lane = 16 * transform + butterfly): the 12 + 12 transforms of the constrained
// filter update take six such rounds.  Every butterfly performs the reference's float operations
// in the reference's order and every cross-bin sum (band averages, PSD sums, partition energies)
// is accumulated by one lane in the reference's order, so spectra, the adaptive filter and the
// NLP state are bit-identical to oracle/aec_oracle.c; only powf / cosf / sinf are evaluated in
// fp64 and rounded (libm's float versions are not correctly rounded everywhere).
//
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "aec_estimator.h"
#include "ns_device.h"  // lean fp64 pow / sincos shared with the NS kernels
#include "pk_f32.h"     // complex arithmetic as packed f32

using namespace aspaec;
using namespace asppk;

#ifndef AEC_FILTERFAR_FIRST
#define AEC_FILTERFAR_FIRST 1  // FilterFar ahead of the near FFT (0: behind it, its row loads in flight under the transform)
#endif

namespace {

// Per-bin loops: trip 0 puts bin q on lane q, trip 1 is bin 64 computed by every lane (uniform
// values, identical stores), so both trips are straight-line code whose loads overlap.
#ifndef AEC_CARRY
#define AEC_CARRY 1  // a block's filter update also accumulates the next block's FilterFar (0: every block loads its own)
#endif
#if AEC_CARRY && !AEC_FILTERFAR_FIRST
#error "AEC_CARRY needs AEC_FILTERFAR_FIRST (the other order recomputes FilterFar in every block)"
#endif
#ifndef AEC_TRIPS
#define AEC_TRIPS 2  // (1: timing probe only -- bin 64 is then not computed and the results are wrong)
#endif
#define BINS_2TRIPS _Pragma("unroll") for (int t_ = 0, bin = lane; t_ < AEC_TRIPS; ++t_, bin = 64)

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------ LDS map
constexpr int kTileStride = 80;                 // float2 per transform tile: 64, or 79 in the padded
                                                // order the passes use between them (ph below); 80 = 16 mod 32 keeps the two tiles of a
                                                // 32-lane group on opposite halves of the 64 banks
constexpr int kLdsTile = 0;                     // 4 tiles: 4 * 80 float2 = 640 floats
constexpr int kLdsRows = 4 * kTileStride * 2;   // per-bin rows of 66 floats
enum LRow { L_XFR = 0, L_XFI, L_DFR, L_DFI, L_YFR, L_YFI, L_EFR, L_EFI, L_XPOW, L_XWR, L_XWI,
            L_COHDE, L_COHXD, L_HNL, L_T0, L_T1, L_DWR, L_DWI, L_EWR, L_EWI, L_NROWS };
constexpr int kLRow = 66;
constexpr int kLdsDbuf = kLdsRows + L_NROWS * kLRow;  // 128
constexpr int kLdsEbuf = kLdsDbuf + 128;              // 128
constexpr int kLdsMisc = kLdsEbuf + 128;              // 16: the four NLP sums / 12 partition energies
constexpr int kLdsC64 = kLdsMisc + 16;                // bin 64 of every state row: 124 floats with 12 partitions, the
                                                      // whole padded column (256) with 32
constexpr int lds_c64_len(int np) { return np == kNumPartNormal ? 124 : kC64Len; }
// 12 partitions: 1980 floats = 7 920 B per wave, four workgroups per CU (40 432 B each); 32: 2112 floats, three
constexpr int lds_wave(int np, bool metrics) { return kLdsC64 + lds_c64_len(np) + (metrics ? 4 * kLRow : 0); }
constexpr int lds_met(int np) { return kLdsC64 + lds_c64_len(np); }
constexpr int kLdsWave = lds_wave(kNumPartNormal, false);  // the far-end / transform-seam kernels (tiles only)
constexpr int kPeScratch = 208;  // 32 partitions: the partition energies sit in the pad of the bin-64 column
static_assert(kPeScratch >= AecRows(kNumPartMax).R_COUNT && kPeScratch + kNumPartMax <= kC64Len, "energy scratch");
// metrics mode appends 4 rows of per-bin energy terms (far / near / linear-out / NLP-out) to each wave's region
constexpr int kPeChunk = 12;  // partitions whose per-bin energies overlay the tiles and the dead rows at a time
static_assert(kPeChunk * kLRow <= kLdsRows + 9 * kLRow, "partition-energy rows overlay");

struct SharedTables {
  double exp2_64[64];
  float2 btw[16][4];  // twiddles of radix-4 block B: w1, w2, w3 (cft1st_128 / cftmdl_128), [3] unused
  float w[64], wk3a[16], wk3b[16], hann[68], weight[68], odrive[68];
  uint32_t lcg_a[64], lcg_c[64];
};

// ---------------------------------------------------------------- butterflies
__device__ __forceinline__ f32x2 pk(float2 v) { return f32x2{v.x, v.y}; }
__device__ __forceinline__ float2 unpk(f32x2 v) { return make_float2(v.x, v.y); }

// Radix-4 butterfly of cft1st_128 / cftmdl_128 for block B of a pass (aec_rdft.c:201-444):
// B = 0 no twiddles, B = 1 the w[2] block, B = 2u / 2u + 1 general.
__device__ __forceinline__ void bfly(float2& e0, float2& e1, float2& e2, float2& e3, int B,
                                     const SharedTables& T) {
  // Branch-free: the lanes of a pass hold different blocks B, so the three forms of the reference
  // (no twiddles, the w[2] block, general) would otherwise run one after the other under divergent
  // branches.  The general form is evaluated for every lane (B = 0 multiplies by (1, 0): harmless),
  // the plain form is its own intermediate values, the w[2] form costs seven more packed operations;
  // the lane then keeps the form of its block.  Every kept value comes from the reference's own
  // operations (complex values as register pairs, packed f32: pk_f32.h).
  const f32x2 a0 = pk(e0), a1 = pk(e1), a2 = pk(e2), a3 = pk(e3);
  const f32x2 x0 = a0 + a1, x1 = a0 - a1, x2 = a2 + a3, x3 = a2 - a3;
  e0 = unpk(x0 + x2);
  const f32x2 w1 = pk(T.btw[B][0]), w2 = pk(T.btw[B][1]), w3 = pk(T.btw[B][2]);
  // plain (B == 0) = the general form's operands
  const f32x2 d0 = x0 - x2;
  const f32x2 y1 = add_swap_sub_lo(x1, x3);  // {x1r - x3i, x1i + x3r}
  const f32x2 y3 = add_swap_sub_hi(x1, x3);  // {x1r + x3i, x1i - x3r}
  // general: {wr vr - wi vi, wr vi + wi vr}
  const f32x2 g2 = cmul<false>(w2, d0), g1 = cmul<false>(w1, y1), g3 = cmul<false>(w3, y3);
  // the w[2] block (B == 1)
  const float ws = T.w[2];
  const f32x2 s2 = {x2.y - x0.y, d0.x};
  const f32x2 s1 = diff_sum(y1) * ws;             // ws * {y1r - y1i, y1r + y1i}
  const f32x2 z = swap0_add_sub_hi(x3, x1);       // {x3i + x1r, x3r - x1i}
  const f32x2 s3 = rdiff_sum(z) * ws;             // ws * {zi - zr, zi + zr}
  const bool plain = B == 0, diag = B == 1;
  e2 = unpk(plain ? d0 : (diag ? s2 : g2));
  e1 = unpk(plain ? y1 : (diag ? s1 : g1));
  e3 = unpk(plain ? y3 : (diag ? s3 : g3));
}

// Last pass of cftfsub_128 / cftbsub_128 (aec_rdft.c:446-507).
__device__ __forceinline__ void bfly_last(float2& e0, float2& e1, float2& e2, float2& e3,
                                          bool backward) {
  const f32x2 a0 = pk(e0), a1 = pk(e1), a2 = pk(e2), a3 = pk(e3);
  const f32x2 x2 = a2 + a3, x3 = a2 - a3;
  if (!backward) {
    const f32x2 x0 = a0 + a1, x1 = a0 - a1;
    e0 = unpk(x0 + x2);
    e2 = unpk(x0 - x2);
    e1 = unpk(add_swap_sub_lo(x1, x3));  // {x1r - x3i, x1i + x3r}
    e3 = unpk(add_swap_sub_hi(x1, x3));  // {x1r + x3i, x1i - x3r}
  } else {
    const f32x2 x0 = add_neg_both_hi(a0, a1);  // {e0r + e1r, -e0i - e1i}
    const f32x2 x1 = sub_lo_rsub_hi(a0, a1);   // {e0r - e1r, -e0i + e1i}
    e0 = unpk(add_sub_hi(x0, x2));             // {x0r + x2r, x0i - x2i}
    e2 = unpk(add_sub_lo(x0, x2));             // {x0r - x2r, x0i + x2i}
    e1 = unpk(sub_swap(x1, x3));               // {x1r - x3i, x1i - x3r}
    e3 = unpk(add_swap(x1, x3));               // {x1r + x3i, x1i + x3r}
  }
}

__device__ __forceinline__ int rev6(int x) { return (int)(__brev((unsigned)x) >> 26); }

// Between the passes an element i sits in slot ph(i) = i + i / 4: the stride-4 writes of the first pass
// and the stride-4 / stride-16 accesses of the second then spread over the LDS banks (in natural order
// the first pass's 8-byte writes were 4-way bank conflicts, the second's accesses 2- to 4-way).
__device__ __forceinline__ constexpr int ph(int i) { return i + (i >> 2); }

// The three radix-4 passes of four transforms at once; t = this lane's tile, b = butterfly.
// kAdapt (backward transforms of FilterAdaptation only): the time-domain result leaves the last pass already
// scaled by 2 / 128 in its first half and zeroed in its second (aec_core.c:245-254), instead of a separate
// trip through the tile.
template <bool kAdapt = false>
__device__ __forceinline__ void cft64_quad(float2* t, int b, bool backward, const SharedTables& T) {
  {  // cft1st_128 on the bit-reversed input (bitrv2_128, aec_rdft.c:124-199)
    const int i0 = 4 * b;
    float2 a0 = t[rev6(i0)], a1 = t[rev6(i0 + 1)], a2 = t[rev6(i0 + 2)], a3 = t[rev6(i0 + 3)];
    bfly(a0, a1, a2, a3, b, T);
    wave_fence();  // every lane has read its natural-order inputs before the padded order overwrites them
    const int p0 = 5 * b;  // ph(4 b + k) = 5 b + k
    t[p0] = a0;
    t[p0 + 1] = a1;
    t[p0 + 2] = a2;
    t[p0 + 3] = a3;
  }
  wave_fence();
  {  // cftmdl_128, l = 4 complex: elements 16 g + h + 4 k -> slots 20 g + h + 5 k
    const int p0 = 20 * (b >> 2) + (b & 3);
    float2 a0 = t[p0], a1 = t[p0 + 5], a2 = t[p0 + 10], a3 = t[p0 + 15];
    bfly(a0, a1, a2, a3, b >> 2, T);
    t[p0] = a0;
    t[p0 + 5] = a1;
    t[p0 + 10] = a2;
    t[p0 + 15] = a3;
  }
  wave_fence();
  {  // elements b + 16 k -> slots ph(b) + 20 k; the results go back in natural order
    const int p0 = ph(b);
    float2 a0 = t[p0], a1 = t[p0 + 20], a2 = t[p0 + 40], a3 = t[p0 + 60];
    bfly_last(a0, a1, a2, a3, backward);
    wave_fence();
    if (kAdapt) {  // complex slots 0..31 = samples 0..63 (kept, scaled), 32..63 = samples 64..127 (zeroed)
      const float scale = 2.0f / 128;
      a0.x *= scale;
      a0.y *= scale;
      a1.x *= scale;
      a1.y *= scale;
      a2 = make_float2(0.f, 0.f);
      a3 = make_float2(0.f, 0.f);
    }
    t[b] = a0;
    t[b + 16] = a1;
    t[b + 32] = a2;
    t[b + 48] = a3;
  }
  wave_fence();
}

// aec_rdft_forward_128 (aec_rdft.c:539-547) of the four tiles of this wave, in place.
__device__ __forceinline__ void rdft_fwd_quad(float* wl, int lane, const SharedTables& T) {
  float2* t = reinterpret_cast<float2*>(wl + kLdsTile) + (lane >> 4) * kTileStride;
  const int b = lane & 15;
  const float* c = T.w + 32;
  cft64_quad(t, b, false, T);
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // rftfsub_128 (aec_rdft.c:509-527), pairs (j1, 64 - j1)
    // lane 15 of the second trip (j1 = 32) has no pair: it does the a[0], a[1] step of
    // aec_rdft.c:544-546 instead.  Both forms are evaluated by every lane (clamped index), only the
    // stores differ, so the trip is one straight line with two small masked stores.
    const int j1 = b + 1 + 16 * r;
    const bool pair = j1 < 32;
    const int jc = pair ? j1 : 31, k = 64 - jc;
    const float wkr = 0.5f - c[32 - jc], wki = c[jc];
    float2 aj = t[jc], ak = t[k];
    float2 a0 = t[0];
    {
      const f32x2 x = add_sub_lo(pk(aj), pk(ak));              // {aj.r - ak.r, aj.i + ak.i}
      const f32x2 y = cmul<false>(f32x2{wkr, wki}, x);          // {wkr xr - wki xi, wkr xi + wki xr}
      aj = unpk(pk(aj) - y);
      ak = unpk(add_sub_hi(pk(ak), y));                         // {ak.r + yr, ak.i - yi}
    }
    const float x0 = a0.x - a0.y;
    a0.x += a0.y;
    a0.y = x0;
    if (pair) {
      t[jc] = aj;
      t[k] = ak;
    } else {
      t[0] = a0;
    }
  }
  wave_fence();
}

// aec_rdft_inverse_128 (aec_rdft.c:549-556) of the four tiles, in place (unscaled).
template <bool kAdapt = false>
__device__ __forceinline__ void rdft_inv_quad(float* wl, int lane, const SharedTables& T) {
  float2* t = reinterpret_cast<float2*>(wl + kLdsTile) + (lane >> 4) * kTileStride;
  const int b = lane & 15;
  const float* c = T.w + 32;
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // rftbsub_128 (aec_rdft.c:529-537); lane 15 of the second trip: a[0], a[1], a[65]
    const int j1 = b + 1 + 16 * r;
    const bool pair = j1 < 32;
    const int jc = pair ? j1 : 31, k = 64 - jc;
    const float wkr = 0.5f - c[32 - jc], wki = c[jc];
    float2 aj = t[jc], ak = t[k];
    float2 a0 = t[0], am = t[32];
    {
      const f32x2 x = add_sub_lo(pk(aj), pk(ak));              // {aj.r - ak.r, aj.i + ak.i}
      const f32x2 y = cmul_conj_w(f32x2{wkr, wki}, x);          // {wkr xr + wki xi, wkr xi - wki xr}
      aj = unpk(sub_lo_rsub_hi(pk(aj), y));                     // {aj.r - yr, yi - aj.i}
      ak = unpk(add_sub_hi(y, pk(ak)));                         // {yr + ak.r, yi - ak.i}
    }
    a0.y = 0.5f * (a0.x - a0.y);
    a0.x -= a0.y;
    a0.y = -a0.y;
    am.y = -am.y;  // a[65] = -a[65]
    if (pair) {
      t[jc] = aj;
      t[k] = ak;
    } else {
      t[0] = a0;
      t[32] = am;
    }
  }
  wave_fence();
  cft64_quad<kAdapt>(t, b, true, T);
}

// One round with both directions: the tiles named in inv_tiles take aec_rdft_inverse_128, the others
// aec_rdft_forward_128 (the radix-4 passes are the same code; the real-FFT pre / post steps and the last
// pass's variant go by the lane's tile).  Same operations per tile as the two functions above.
__device__ __forceinline__ void rdft_mixed_quad(float* wl, int lane, const SharedTables& T, int inv_tiles) {
  float2* t = reinterpret_cast<float2*>(wl + kLdsTile) + (lane >> 4) * kTileStride;
  const int b = lane & 15;
  const float* c = T.w + 32;
  const bool inv = ((inv_tiles >> (lane >> 4)) & 1) != 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // rftbsub_128 on the inverse tiles (stores masked)
    const int j1 = b + 1 + 16 * r;
    const bool pair = j1 < 32;
    const int jc = pair ? j1 : 31, k = 64 - jc;
    const float wkr = 0.5f - c[32 - jc], wki = c[jc];
    float2 aj = t[jc], ak = t[k];
    float2 a0 = t[0], am = t[32];
    {
      const f32x2 x = add_sub_lo(pk(aj), pk(ak));
      const f32x2 y = cmul_conj_w(f32x2{wkr, wki}, x);
      aj = unpk(sub_lo_rsub_hi(pk(aj), y));
      ak = unpk(add_sub_hi(y, pk(ak)));
    }
    a0.y = 0.5f * (a0.x - a0.y);
    a0.x -= a0.y;
    a0.y = -a0.y;
    am.y = -am.y;
    if (inv) {
      if (pair) {
        t[jc] = aj;
        t[k] = ak;
      } else {
        t[0] = a0;
        t[32] = am;
      }
    }
  }
  wave_fence();
  cft64_quad(t, b, inv, T);
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // rftfsub_128 on the forward tiles (stores masked)
    const int j1 = b + 1 + 16 * r;
    const bool pair = j1 < 32;
    const int jc = pair ? j1 : 31, k = 64 - jc;
    const float wkr = 0.5f - c[32 - jc], wki = c[jc];
    float2 aj = t[jc], ak = t[k];
    float2 a0 = t[0];
    {
      const f32x2 x = add_sub_lo(pk(aj), pk(ak));
      const f32x2 y = cmul<false>(f32x2{wkr, wki}, x);
      aj = unpk(pk(aj) - y);
      ak = unpk(add_sub_hi(pk(ak), y));
    }
    const float x0 = a0.x - a0.y;
    a0.x += a0.y;
    a0.y = x0;
    if (!inv) {
      if (pair) {
        t[jc] = aj;
        t[k] = ak;
      } else {
        t[0] = a0;
      }
    }
  }
  wave_fence();
}

__device__ __forceinline__ float* lrow(float* wl, int r) { return wl + kLdsRows + r * kLRow; }
__device__ __forceinline__ float2* tile(float* wl, int f) {
  return reinterpret_cast<float2*>(wl + kLdsTile) + f * kTileStride;
}

// Spectrum of tile f -> planar rows (TimeToFrequency reorder / StoreAsComplex, aec_core.c:786-795).
__device__ __forceinline__ void unpack_tile(float* wl, int f, float* re, float* im, int lane) {
  const float2 v = tile(wl, f)[lane];
  if (lane == 0) {
    re[0] = v.x;
    im[0] = 0.f;
    re[64] = v.y;
    im[64] = 0.f;
  } else {
    re[lane] = v.x;
    im[lane] = v.y;
  }
}

__device__ __forceinline__ void stage_tables(SharedTables& S, const AecTables* __restrict__ G) {
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    S.w[i] = G->w[i];
    S.exp2_64[i] = G->exp2_64[i];
    S.lcg_a[i] = G->lcg_a[i];
    S.lcg_c[i] = G->lcg_c[i];
  }
  for (int i = threadIdx.x; i < 16; i += blockDim.x) {
    S.wk3a[i] = G->wk3a[i];
    S.wk3b[i] = G->wk3b[i];
    // twiddles of block B = i (aec_rdft.c:201-444): wk2 = w[k1..], wk1 = w[k2 + 2 odd ..], wk3 from the
    // wk3ri tables; an odd block multiplies by (-wk2i, wk2r)
    const int odd = i & 1, k1 = 2 * (i >> 1), k2 = 2 * k1;
    const float wk2r = G->w[k1], wk2i = G->w[k1 + 1];
    const float* wk3 = odd ? G->wk3b : G->wk3a;
    S.btw[i][0] = make_float2(G->w[k2 + 2 * odd], G->w[k2 + 2 * odd + 1]);
    S.btw[i][1] = odd ? make_float2(-wk2i, wk2r) : make_float2(wk2r, wk2i);
    S.btw[i][2] = make_float2(wk3[k1], wk3[k1 + 1]);
    S.btw[i][3] = make_float2(0.f, 0.f);
  }
  for (int i = threadIdx.x; i < 68; i += blockDim.x) {
    S.hann[i] = G->hann[i];
    S.weight[i] = G->weight[i];
    S.odrive[i] = G->odrive[i];
  }
  __syncthreads();
}

__device__ __forceinline__ int ring_idx(int pos, int i, int count) {
  int p = pos + i;
  while (p >= count) p -= count;
  return p;
}

// The stream's state block through a buffer resource: one descriptor in SGPRs, the row offset as the
// scalar offset, the lane's dword as the only vector offset.  (With flat global addressing the compiler
// kept a 64-bit VGPR address per state row alive across the whole block: 100 registers.)
// AUX: the cache-policy operand of every access through the descriptor: 0, or kSc1 in the hand-off build (below).
constexpr int kSc1 = 16;
template <int AUX>
struct StateBufT {
  __amdgpu_buffer_rsrc_t r;
};
using StateBuf = StateBufT<0>;
template <int AUX = 0>
__device__ __forceinline__ StateBufT<AUX> state_buf(const float* st, int dwords) {
  StateBufT<AUX> b;
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st), 0, dwords * 4, 0x00020000);
  return b;
}
// dword `uni + vec` of the block: uni wave-uniform, vec per lane
template <int AUX>
__device__ __forceinline__ float sld(const StateBufT<AUX>& b, int uni, int vec) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, vec * 4, uni * 4, AUX));
}
template <int AUX>
__device__ __forceinline__ void sst(const StateBufT<AUX>& b, int uni, int vec, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, vec * 4, uni * 4, AUX);
}

// ---- the hand-off build (FLOW): consecutive frame steps overlap on the chip (the scheme of ns_kernels1.hip).
// One launch carries M consecutive WebRtcAec_Process calls (each with the BufferFarend call before it) of the
// whole batch: blockIdx.y is the step, its descriptor steps[blockIdx.y] sits in device memory.  Workgroups are
// dispatched in linear order (x fastest), so the wave that takes stream s in step j + 1 may wait for the one
// that has stream s in step j: it is resident or done.  The hand-off is a word in memory: seq[s] = k + 1 stored
// by the wave that finished stream s of step k, its stores drained first; polled by the wave of step k + 1
// before its first access to the stream's state or far-ring slots.  Every access to what a stream's steps hand
// each other -- the state block and the far ring -- is an sc1 access (write-through stores, L1-bypassing loads;
// MI355X_MICROARCH.md, visibility).  Frames in / out and the tables are plain.  The wait is bounded: a wave that
// gives up sets the abort word, which every later wait sees (aec_api.hip reports it).
struct AecFlowArgs {
  const AecFlowStep* steps;  // [gridDim.y]
  unsigned* seq;             // [num_streams]: hand-off steps stream s has completed
  unsigned* abort_w;         // != 0: a wait timed out (1 + stream)
  unsigned want;             // blockIdx.y == 0 is step `want` of every stream
  DelayBlock* est;           // delay logging: the streams' estimator blocks (their mean spectra are kept here) and the
  unsigned* bits;            // launch's binary spectra [stream][kFlowBitsBlocks][far, near]; nullptr: off
};
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ bool flow_wait(const AecFlowArgs& fa, unsigned want, int stream, int lane) {
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
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the state loads below the poll
  return true;
}

// The same into rows `re` / `im` (dword offsets) of a block behind a descriptor.
template <int AUX>
__device__ __forceinline__ void unpack_tile_buf(float* wl, int f, const StateBufT<AUX>& b, int re, int im, int lane) {
  const float2 v = tile(wl, f)[lane];
  if (lane == 0) {
    sst(b, re, 0, v.x);
    sst(b, im, 0, 0.f);
    sst(b, re, 64, v.y);
    sst(b, im, 64, 0.f);
  } else {
    sst(b, re, lane, v.x);
    sst(b, im, lane, v.y);
  }
}

// ------------------------------------------------------------------ far end
// The far-end work of one WebRtcAec_BufferFarend call for this wave's stream.
// Delay logging in the hand-off build (aec_core.c:1191-1203): the block's two binary spectra are formed here, where
// |X|^2 and |D|^2 of bins 0..63 are in registers (bands 12..43 are all the estimator looks at), against the mean
// spectra in the stream's estimator block; the rest of the estimator runs once per launch (aec_delay_bits_kernel).
// The mean spectra pass from one step's wave to the next like the state block: sc1 accesses.
struct EstOffsets {
  static constexpr int oFar = __builtin_offsetof(AspAecDelayState, mean_far_spectrum) / 4;
  static constexpr int oNear = __builtin_offsetof(AspAecDelayState, mean_near_spectrum) / 4;
  static constexpr int oFarInit = __builtin_offsetof(AspAecDelayState, far_spectrum_initialized) / 4;
  static constexpr int oNearInit = __builtin_offsetof(AspAecDelayState, near_spectrum_initialized) / 4;
};
__device__ __forceinline__ StateBufT<kSc1> est_buf(const DelayBlock* est) {
  return state_buf<kSc1>(reinterpret_cast<const float*>(&est->s), (int)(sizeof(AspAecDelayState) / 4));
}
// thr_far / thr_near: the lane's band means; the flags as float bits (all read at the top of the block, parked in LDS)
__device__ __forceinline__ void flow_binary_spectra(const DelayBlock* est, unsigned* bits_out, float far_pow, float near_pow,
                                                    float thr_far, float thr_near, float far_init_f, float near_init_f,
                                                    int lane) {
  const StateBufT<kSc1> eb = est_buf(est);
  int far_init = __builtin_amdgcn_readfirstlane(__float_as_int(far_init_f));
  int near_init = __builtin_amdgcn_readfirstlane(__float_as_int(near_init_f));
  const unsigned bfar = binary_spectrum(sqrtf(far_pow), thr_far, far_init, lane);
  const unsigned bnear = binary_spectrum(sqrtf(near_pow), thr_near, near_init, lane);
  sst(eb, EstOffsets::oFar, lane, thr_far);
  sst(eb, EstOffsets::oNear, lane, thr_near);
  if (lane == 0) {
    sst(eb, EstOffsets::oFarInit, 0, __int_as_float(far_init));
    sst(eb, EstOffsets::oNearInit, 0, __int_as_float(near_init));
    bits_out[0] = bfar;  // read by a later launch
    bits_out[1] = bnear;
  }
}

template <bool FLOW = false, class OPS = FarOps>
__device__ __forceinline__ void farend_work(float* __restrict__ st, float* __restrict__ far_ring,
                                            float* __restrict__ wl, const SharedTables& T,
                                            const float* __restrict__ farend, int num_streams,
                                            int stream, const OPS& ops, int lane, int state_dwords = 0) {
  if constexpr (FLOW) {
    // the hand-off build: far_pre and the far-ring slots through descriptors, every access sc1
    const StateBufT<kSc1> sb = state_buf<kSc1>(st, state_dwords);
    for (int i = lane; i < ops.n; i += 64) sst(sb, kOffPre, ring_idx(ops.wpos, i, kPreLen), farend[(size_t)stream * ops.n + i]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int p = 0; p < ops.nparts; p += 2) {
      const int np = ops.nparts - p < 2 ? ops.nparts - p : 2;  // wave-uniform
      for (int q = 0; q < np; ++q) {
        float* t0 = reinterpret_cast<float*>(tile(wl, 2 * q));
        float* t1 = reinterpret_cast<float*>(tile(wl, 2 * q + 1));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = lane + 64 * h;
          const float x = sld(sb, kOffPre, ring_idx(ops.rpos[p + q], i, kPreLen));
          t0[i] = x;
          t1[i] = x * (h == 0 ? T.hann[i] : T.hann[128 - i]);
        }
      }
      wave_fence();
      rdft_fwd_quad(wl, lane, T);
      for (int q = 0; q < np; ++q) {
        const StateBufT<kSc1> fb =
            state_buf<kSc1>(far_ring + ((size_t)ops.slot[p + q] * num_streams + stream) * kFarSlotDwords, kFarSlotDwords);
        unpack_tile_buf(wl, 2 * q, fb, 0, kRow, lane);
        unpack_tile_buf(wl, 2 * q + 1, fb, 2 * kRow, 3 * kRow, lane);
      }
      wave_fence();
    }
    return;
  }
  float* pre = st + kOffPre;
  for (int i = lane; i < ops.n; i += 64) pre[ring_idx(ops.wpos, i, kPreLen)] = farend[(size_t)stream * ops.n + i];
  // the partitions below read samples other lanes of this wave just wrote: order them (same CU,
  // same L1: workgroup scope; agent scope would flush the XCD L2)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int p = 0; p < ops.nparts; p += 2) {
    // two partitions per transform round: 128 samples of far_pre -> tiles 0 / 2 (plain) and 1 / 3 (windowed,
    // aec_core.c:779-784)
    const int np = ops.nparts - p < 2 ? ops.nparts - p : 2;  // wave-uniform
    for (int q = 0; q < np; ++q) {
      float* t0 = reinterpret_cast<float*>(tile(wl, 2 * q));
      float* t1 = reinterpret_cast<float*>(tile(wl, 2 * q + 1));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        const float x = pre[ring_idx(ops.rpos[p + q], i, kPreLen)];
        t0[i] = x;
        t1[i] = x * (h == 0 ? T.hann[i] : T.hann[128 - i]);
      }
    }
    wave_fence();
    rdft_fwd_quad(wl, lane, T);
    for (int q = 0; q < np; ++q) {
      float* slot = far_ring + ((size_t)ops.slot[p + q] * num_streams + stream) * kFarSlotDwords;
      unpack_tile(wl, 2 * q, slot, slot + kRow, lane);
      unpack_tile(wl, 2 * q + 1, slot + 2 * kRow, slot + 3 * kRow, lane);
    }
    wave_fence();
  }
}

// WebRtcAec_BufferFarend for every stream: append to far_pre, then transform the partitions the
// host scheduled (plain and sqrt-Hann windowed) into their far-ring slots.
__global__ __launch_bounds__(256) void aec_farend_kernel(int stream0, int stream_end, float* __restrict__ state,
                                                         float* __restrict__ far_ring,
                                                         const AecTables* __restrict__ G,
                                                         const float* __restrict__ farend,
                                                         int num_streams, FarOps ops, int state_dwords) {
  __shared__ SharedTables T;
  __shared__ float lds[4 * kLdsWave];
  stage_tables(T, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: the stream's state block and LDS region get scalar bases
  const int stream = stream0 + blockIdx.x * 4 + wave;  // this launch covers streams stream0 .. stream_end - 1
  if (stream >= stream_end) return;
  float* wl = lds + wave * kLdsWave;
  float* st = state + (size_t)stream * state_dwords;
  farend_work(st, far_ring, wl, T, farend, num_streams, stream, ops, lane);
}

// The NLP's powf / cosf / sinf are evaluated as (float)f((double)x) by the lean fp64 forms of
// ns_device.h (identical to the plain ocml forms for every argument, tests/test_aec_gpu.py); glibc's
// float versions (<= 0.52 / 0.56 ulp) differ from that in rare last-place cases.

// Out of line on purpose: inlined, their fp64 polynomials push process_block into spilling.
__device__ __attribute__((noinline)) float nlp_pow(float x, float y, const double* __restrict__ t64) {
  return aspns_dev::pow_f32_via_f64(x, y, t64);
}
__device__ __attribute__((noinline)) float2 nlp_sincos(float x) {
  float2 r;
  aspns_dev::sincos_f32_via_f64(x, r.x, r.y);
  return r;
}

// The high band of a 32 kHz stream (aec_core.c:1032-1067, 501-545, 451-459).  Out of line on
// purpose: it keeps the one-band path's register allocation untouched (its LDS accesses become
// generic here, which only the two-band path pays for).
__device__ __attribute__((noinline)) void high_band_block(float* __restrict__ st, float* __restrict__ wl,
                                                          const SharedTables& T, int near_rpos,
                                                          int out_wpos, int lane) {
  float* misc = wl + kLdsMisc;
  float* HNL = lrow(wl, L_HNL);
  float* DFR = lrow(wl, L_DFR);
  float* DFI = lrow(wl, L_DFI);
  float* YFR = lrow(wl, L_YFR);
  float* YFI = lrow(wl, L_YFI);
  const float scale = 2.0f / 128;
    // ---- high band (aec_core.c:1032-1067, 501-545, 451-459): the three averages are summed by
    // one lane each in the reference's order
    wave_fence();
    if (lane < 3) {
      const float* src = lane == 0 ? HNL : lane == 1 ? DFR : DFI;
      float acc = 0.f;
#pragma unroll
      for (int j = 32; j < 65; ++j)
        if (j < 64 || lane > 0) acc += src[j];
      misc[8 + lane] = acc;
    }
    wave_fence();
    const float nlpGainHband = misc[8] / 32.0f;
    const float noiseAvg = misc[9] / 33.0f, tmpAvg = misc[10] / 33.0f;
    {
      // comfortNoiseHband packed for the inverse transform: fft[0] = cn[0].re (0), fft[1] = cn[64].re
      float2 v;
      v.x = lane == 0 ? 0.f : tmpAvg * (noiseAvg * YFR[lane]);
      v.y = lane == 0 ? tmpAvg * (noiseAvg * YFR[64]) : tmpAvg * (-noiseAvg * YFI[lane]);
      tile(wl, 0)[lane] = v;
    }
    wave_fence();
    rdft_inv_quad(wl, lane, T);
    {
      const float cn = reinterpret_cast<const float*>(tile(wl, 0))[lane] * scale;
      const float nearH = st[kOffNearFrH + ring_idx(near_rpos, lane, kFrBufLen)];
      float dtmp = st[kOffDBufH + lane];
      dtmp = dtmp * nlpGainHband;
      dtmp += 0.4f * cn;  // cnScaleHband, aec_core.c:45-46
      st[kOffOutFrH + ring_idx(out_wpos, lane, kFrBufLen)] =
          dtmp > 32767.f ? 32767.f : (dtmp < -32768.f ? -32768.f : dtmp);
      st[kOffDBufH + lane] = nearH;
    }
    wave_fence();
}

// UpdateLevel x 4 + UpdateMetrics (aec_core.c:585-770) from the per-bin energy terms the block
// left in the wave's kLdsMet rows: lanes 0..3 each sum one level in the reference's order (bin 0, bin 64,
// bins 1..63) and update its PowerLevel; lane 0 then updates the ERL / A_NLP / ERLE statistics.
// `met` is this stream's AspAecMetricsState image (include/asp_aec.h).
__device__ __attribute__((noinline)) void metrics_block(float* __restrict__ met, float* __restrict__ metrows,
                                                        int lane, int echoState) {
  constexpr int subCountLen = 4, countLen = 50;  // aec_core.c:47-48
  int32_t* meti = reinterpret_cast<int32_t*>(met);
  wave_fence();
  float minlevel = 0.f, averagelevel = 0.f;
  int frcounter = 0, sfrcounter = 0;
  if (lane < 4) {
    const float* src = metrows + lane * kLRow;
    float energy = src[0];
    energy += src[64];
#pragma unroll
    for (int k = 1; k < 64; ++k) energy += src[k];
    energy /= 128;
    float* L = met + kMetLevel * lane;  // sfrsum sfrcounter framelevel frsum frcounter minlevel averagelevel
    int32_t* Li = meti + kMetLevel * lane;
    float sfrsum = L[0] + energy;
    sfrcounter = Li[1] + 1;
    frcounter = Li[4];
    minlevel = L[5];
    averagelevel = L[6];
    if (sfrcounter > subCountLen) {
      const float framelevel = sfrsum / (subCountLen * kPartLen);
      L[2] = framelevel;
      sfrsum = 0;
      sfrcounter = 0;
      if (framelevel > 0) {
        if (framelevel < minlevel) {
          minlevel = framelevel;
        } else {
          minlevel *= (1 + 0.001f);
        }
      }
      frcounter++;
      float frsum = L[3] + framelevel;
      if (frcounter > countLen) {
        averagelevel = frsum / countLen;
        frsum = 0;
        frcounter = 0;
      }
      L[3] = frsum;
      Li[4] = frcounter;
      L[5] = minlevel;
      L[6] = averagelevel;
    }
    L[0] = sfrsum;
    Li[1] = sfrcounter;
  }
  const float near_avg = __shfl(averagelevel, 1), near_min = __shfl(minlevel, 1);
  const float lin_avg = __shfl(averagelevel, 2), lin_min = __shfl(minlevel, 2);
  const float nlp_avg = __shfl(averagelevel, 3), nlp_min = __shfl(minlevel, 3);
  if (lane == 0) {  // UpdateMetrics, aec_core.c:644-770 (lane 0 holds the far level)
    int stateCounter = meti[kMetStateCounter];
    if (echoState) stateCounter++;
    if (frcounter == 0) {
      const float actThreshold = minlevel < 300000.0f ? 40.0f : 8.0f;
      if ((stateCounter > (0.5f * countLen * subCountLen)) && (sfrcounter == 0) &&
          (averagelevel > (actThreshold * minlevel))) {
        const float safety = 0.99995f;
        const float echo = near_avg - safety * near_min;
        auto add = [&](int which, float instant, float dtmp) {
          float* S = met + kMetStats + kMetStat * which;  // instant average min max sum hisum himean counter hicounter
          int32_t* Si = meti + kMetStats + kMetStat * which;
          S[0] = instant;
          if (dtmp > S[3]) S[3] = dtmp;
          if (dtmp < S[2]) S[2] = dtmp;
          const int counter = Si[7] + 1;
          Si[7] = counter;
          const float sum = S[4] + dtmp;
          S[4] = sum;
          const float average = sum / counter;
          S[1] = average;
          if (dtmp > average) {
            const int hicounter = Si[8] + 1;
            Si[8] = hicounter;
            const float hisum = S[5] + dtmp;
            S[5] = hisum;
            S[6] = hisum / hicounter;
          }
        };
        float dtmp = 10 * (float)log10((double)(averagelevel / near_avg + 1e-10f));
        add(0, dtmp, dtmp);  // ERL
        dtmp = 10 * (float)log10((double)(near_avg / (2 * lin_avg) + 1e-10f));
        float suppressedEcho = 2 * (lin_avg - safety * lin_min);
        float dtmp2 = 10 * (float)log10((double)(echo / suppressedEcho + 1e-10f));
        add(2, dtmp2, dtmp);  // A_NLP
        suppressedEcho = 2 * (nlp_avg - safety * nlp_min);
        dtmp2 = 10 * (float)log10((double)(echo / suppressedEcho + 1e-10f));
        add(1, dtmp2, dtmp2);  // ERLE
      }
      stateCounter = 0;
    }
    meti[kMetStateCounter] = stateCounter;
  }
  wave_fence();
}

// One ProcessBlock + NonLinearProcessing for this wave's stream.  kMetrics: metricsMode builds of
// the kernel also gather the echo metrics (a separate instantiation keeps the plain one's
// register allocation).
// n / d, correctly rounded: the lean Newton form of ns_device.h where it is exact (normal divisor and
// quotient, far from the range limits), the IEEE expansion otherwise (rare; one branch per use).
__device__ __forceinline__ float fdiv_aec(float n, float d) {
  float q = aspns_dev::fdiv(n, d);
  const float aq = fabsf(q);
  if (__builtin_expect(!(d >= 1e-18f && d <= 1e18f && ((aq >= 1e-25f && aq <= 1e25f) || n == 0.0f)), 0)) q = n / d;
  return q;
}

// v_writelane_b32: the (wave-uniform) value goes into lane K of the row
template <int K>
__device__ __forceinline__ float set_lane(float row, int bits) {
  int r = __float_as_int(row);
  const int u = __builtin_amdgcn_readfirstlane(bits);
  asm("v_writelane_b32 %0, %1, %2" : "+v"(r) : "s"(u), "n"(K));
  return __int_as_float(r);
}

// NP = 12 partitions, or 32: the extended filter (aec_core_internal.h:23-25, WebRtcAec_enable_delay_correction)
// BITS: the block's two binary spectra are formed here (delay logging in the hand-off build; the fused delay-agnostic
// form), spec_out = where its two words go
template <bool kMetrics, int NP, bool FLOW = false, bool BITS = FLOW, class BLOCKOP = BlockOp>
__device__ __forceinline__ void process_block(float* __restrict__ st, float* __restrict__ wl,
                                              const float* __restrict__ far_slot,
                                              const SharedTables& T, const BLOCKOP& op, int mult,
                                              int nlp_mode, float mu, float error_threshold,
                                              int lane, const double* __restrict__ exp2_global,
                                              int num_high, float* __restrict__ met,
                                              unsigned long long* stamps, float* __restrict__ spec_out,
                                              bool carry_in, const float* __restrict__ next_slot, int next_xf_pos,
                                              const DelayBlock* est = nullptr) {
  // carry_in: the echo estimate's spectrum of this block is already in the YFR / YFI rows (the previous block of
  // this call accumulated it during its filter update); next_slot != nullptr: this block does the same for the
  // next one, whose far spectrum sits in next_slot and whose xfBufBlockPos is next_xf_pos.
  // diagnostic phase stamps (never enabled by the product entry points)
#define AEC_STAMP(k) \
  if (stamps != nullptr) stamps[k] = __builtin_amdgcn_s_memtime();
  AEC_STAMP(0)
  asm volatile("" : "+v"(lane));  // lane masks are recomputed per block instead of living in SGPR pairs across the block loop
  constexpr AecRows RW(NP);
  constexpr int kNumPart = NP, R_XF_IM = RW.R_XF_IM, R_WF_RE = RW.R_WF_RE, R_WF_IM = RW.R_WF_IM, R_XFW = RW.R_XFW,
                R_COUNT = RW.R_COUNT;
  constexpr bool kExtended = NP == kNumPartMax;
  constexpr int kC64Chunks = (R_COUNT + 63) / 64, kC64Lds = lds_c64_len(NP);
  static_assert(R_COUNT <= kC64Lds, "bin-64 column in LDS");
  constexpr int kAux = FLOW ? kSc1 : 0;
  const StateBufT<kAux> sb = state_buf<kAux>(st, RW.state_dwords);
  // the stream's 32 scalars: one row load, lane k holds scalar k (wave-uniform values read with v_readlane),
  // one row store at the end -- a scalar address apiece had cost an SGPR pair each across the block loop
  float scrow = sld(sb, kOffScalars, lane & 31);
#define SCI(k) __builtin_amdgcn_readlane(__float_as_int(scrow), (k))
#define SCF(k) __int_as_float(SCI(k))
#define SC_SETI(k, val) scrow = set_lane<(k)>(scrow, (int)(val))
#define SC_SETF(k, val) SC_SETI(k, __float_as_int(val))
  float* dbuf = wl + kLdsDbuf;
  float* ebuf = wl + kLdsEbuf;
  float* misc = wl + kLdsMisc;
  float* XFR = lrow(wl, L_XFR);
  float* XFI = lrow(wl, L_XFI);
  float* DFR = lrow(wl, L_DFR);
  float* DFI = lrow(wl, L_DFI);
  float* YFR = lrow(wl, L_YFR);
  float* YFI = lrow(wl, L_YFI);
  float* EFR = lrow(wl, L_EFR);
  float* EFI = lrow(wl, L_EFI);
  float* XPW = lrow(wl, L_XPOW);
  float* XWR = lrow(wl, L_XWR);
  float* XWI = lrow(wl, L_XWI);
  float* COHDE = lrow(wl, L_COHDE);
  float* COHXD = lrow(wl, L_COHXD);
  float* HNL = lrow(wl, L_HNL);
  float* T0 = lrow(wl, L_T0);
  float* T1 = lrow(wl, L_T1);
  float* DWR = lrow(wl, L_DWR);  // windowed near spectrum (NLP)
  float* DWI = lrow(wl, L_DWI);
  float* EWR = lrow(wl, L_EWR);  // windowed error spectrum (NLP)
  float* EWI = lrow(wl, L_EWI);
  [[maybe_unused]] float* MET = wl + lds_met(NP);  // [4][kLRow], metrics mode only
  const float scale = 2.0f / 128;
  // State rows: trip 0 (bin = lane) goes to HBM, trip 1 (bin 64) to a per-wave LDS copy of the
  // bin-64 column that is gathered once per block and scattered back at its end.
  float* c64 = wl + kLdsC64;
#define ROWO(r) (kOffRows + (r) * kRowS)
#define ROW_LD(r) (t_ == 0 ? sld(sb, ROWO(r), lane) : c64[r])
#define ROW_ST(r, v)                         \
  do {                                       \
    if (t_ == 0) {                           \
      sst(sb, ROWO(r), lane, (v));       \
    } else {                                 \
      c64[r] = (v);                          \
    }                                        \
  } while (0)
  // delay logging in the hand-off build: the band means of the binary spectra, asked for first and parked in the
  // (still dead) error-spectrum rows with the first LDS writes below, so that the power phase finds them in LDS
  [[maybe_unused]] float est_tf = 0.f, est_tn = 0.f, est_fi = 0.f, est_ni = 0.f;
  if constexpr (BITS) {
    if (spec_out != nullptr) {
      const StateBufT<kSc1> eb = est_buf(est);
      est_tf = sld(eb, EstOffsets::oFar, lane);
      est_tn = sld(eb, EstOffsets::oNear, lane);
      est_fi = sld(eb, EstOffsets::oFarInit, 0);
      est_ni = sld(eb, EstOffsets::oNearInit, 0);
    }
  }
  // loads return in order: the few the first FFT round waits for go first
  float c64_in[kC64Chunks];
#pragma unroll
  for (int k = 0; k < kC64Chunks; ++k) c64_in[k] = sld(sb, kOffC64 + 64 * k, lane);  // the column is padded to 256
  float fs_lane[4], fs_64[4];  // this block's far spectra (plain re/im, windowed re/im)
  if constexpr (FLOW) {
    const StateBufT<kSc1> fb = state_buf<kSc1>(far_slot, kFarSlotDwords);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fs_lane[k] = sld(fb, k * kRow, lane);
      fs_64[k] = sld(fb, k * kRow + 64, 0);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fs_lane[k] = far_slot[k * kRow + lane];
      fs_64[k] = far_slot[k * kRow + 64];
    }
  }
  // ---- near block (aec_core.c:1114-1124) and the far spectra of this block (:1137, 888-891)
  const float ne = sld(sb, kOffNearFr, ring_idx(op.near_rpos, lane, kFrBufLen));
  dbuf[lane] = sld(sb, kOffDBuf, lane);
  dbuf[64 + lane] = ne;
  ebuf[lane] = sld(sb, kOffEBuf, lane);
  if constexpr (BITS) {
    if (spec_out != nullptr) {
      EFR[lane] = est_tf;
      EFI[lane] = est_tn;
      misc[6] = est_fi;
      misc[7] = est_ni;
    }
  }
#pragma unroll
  for (int k = 0; k < kC64Chunks; ++k)
    if (64 * (k + 1) <= kC64Lds || 64 * k + lane < kC64Lds) c64[64 * k + lane] = c64_in[k];
  // The lane's bins of the far-spectrum history and of the filter, for FilterFar only (logical partition
  // order; partition 0 is this block's spectrum).  FilterFar depends on nothing but these, so it runs
  // ahead of the near FFT and the 46 registers are free again before the ten FFT rounds of the block;
  // FilterAdaptation streams its partitions again, four at a time.
  // (the 32 partitions of the extended filter do not fit in registers: their FilterFar loads row by row)
  constexpr int kFarRegs = kExtended ? 1 : kNumPart, kFarUnroll = kExtended ? 8 : kNumPart;
  float xr[kFarRegs], xi[kFarRegs], wr[kFarRegs], wi[kFarRegs];
  float p_xpow, p_dpow, p_dmin, p_dinit, p_outbuf;
  // rows of the power / noise-floor update and the overlap-add tail: in flight during the first FFT
#define AEC_LOAD_POWER_ROWS()                  \
  p_xpow = sld(sb, ROWO(R_XPOW), lane);        \
  p_dpow = sld(sb, ROWO(R_DPOW), lane);        \
  p_dmin = sld(sb, ROWO(R_DMINPOW), lane);     \
  p_dinit = sld(sb, ROWO(R_DINITMINPOW), lane); \
  p_outbuf = sld(sb, kOffOutBuf, lane);        \
  wave_fence();
  // the far spectra of this block (aec_core.c:1137, 888-891)
#define AEC_FAR_SPECTRA()                                                                      \
  BINS_2TRIPS {                                                                                \
    XFR[bin] = t_ == 0 ? fs_lane[0] : fs_64[0];                                                \
    XFI[bin] = t_ == 0 ? fs_lane[1] : fs_64[1];                                                \
    const float xwr = t_ == 0 ? fs_lane[2] : fs_64[2], xwi = t_ == 0 ? fs_lane[3] : fs_64[3];  \
    XWR[bin] = xwr;                                                                            \
    XWI[bin] = xwi;                                                                            \
    ROW_ST((R_XFW + 2 * op.xfw_head), xwr);                                                    \
    ROW_ST((R_XFW + 2 * op.xfw_head + 1), xwi);                                                \
  }                                                                                            \
  wave_fence();
#if AEC_FILTERFAR_FIRST
  if (carry_in) {  // the echo estimate's spectrum waits in YFR / YFI (the previous block's filter update left it)
    AEC_LOAD_POWER_ROWS()
    AEC_FAR_SPECTRA()
  } else {
#pragma unroll
    for (int i = 0; i < kFarRegs; ++i) {
      int px = i + op.xf_pos;
      if (px >= kNumPart) px -= kNumPart;
      if (i > 0) {
        xr[i] = sld(sb, ROWO((R_XF_RE + px)), lane);
        xi[i] = sld(sb, ROWO((R_XF_IM + px)), lane);
      }
      wr[i] = sld(sb, ROWO((R_WF_RE + i)), lane);
      wi[i] = sld(sb, ROWO((R_WF_IM + i)), lane);
    }
    AEC_LOAD_POWER_ROWS()
    xr[0] = fs_lane[0];
    xi[0] = fs_lane[1];
    AEC_FAR_SPECTRA()
    // ---- FilterFar (aec_core.c:147-169): partitions in order, per bin
    BINS_2TRIPS {
      float yr = 0.f, yi = 0.f;
#pragma unroll kFarUnroll
      for (int i = 0; i < kNumPart; ++i) {
        int px = i + op.xf_pos;
        if (px >= kNumPart) px -= kNumPart;
        float ar, ai, br, bi;
        if constexpr (kExtended) {
          ar = i == 0 ? XFR[bin] : ROW_LD((R_XF_RE + px));
          ai = i == 0 ? XFI[bin] : ROW_LD((R_XF_IM + px));
          br = ROW_LD((R_WF_RE + i));
          bi = ROW_LD((R_WF_IM + i));
        } else {
          ar = t_ == 0 ? xr[i] : (i == 0 ? XFR[64] : c64[R_XF_RE + px]);
          ai = t_ == 0 ? xi[i] : (i == 0 ? XFI[64] : c64[R_XF_IM + px]);
          br = t_ == 0 ? wr[i] : c64[R_WF_RE + i];
          bi = t_ == 0 ? wi[i] : c64[R_WF_IM + i];
        }
        yr += ar * br - ai * bi;
        yi += ar * bi + ai * br;
      }
      YFR[bin] = yr;
      YFI[bin] = yi;
    }
    wave_fence();
  }
#else
#pragma unroll
  for (int i = 0; i < kFarRegs; ++i) {
    int px = i + op.xf_pos;
    if (px >= kNumPart) px -= kNumPart;
    if (i > 0) {
      xr[i] = sld(sb, ROWO((R_XF_RE + px)), lane);
      xi[i] = sld(sb, ROWO((R_XF_IM + px)), lane);
    }
    wr[i] = sld(sb, ROWO((R_WF_RE + i)), lane);
    wi[i] = sld(sb, ROWO((R_WF_IM + i)), lane);
  }
  AEC_LOAD_POWER_ROWS()
#endif
  AEC_STAMP(1)
  // ---- near fft (aec_core.c:1140-1141)
  {
    // plain (aec_core.c:1140) and sqrt-Hann windowed (aec_core.c:428-431) near spectra side by side
    const float2 v = {dbuf[2 * lane], dbuf[2 * lane + 1]};
    tile(wl, 0)[lane] = v;
    float* t1 = reinterpret_cast<float*>(tile(wl, 1));
    t1[lane] = dbuf[lane] * T.hann[lane];
    t1[64 + lane] = dbuf[64 + lane] * T.hann[64 - lane];
#if AEC_FILTERFAR_FIRST
    // the echo estimate's spectrum (FilterFar ran above) rides in the same round, third tile, inverse
    // (aec_core.c:1222-1238): one transform round fewer per block
    float2 vy;
    vy.x = YFR[lane];
    vy.y = lane == 0 ? YFR[64] : YFI[lane];
    tile(wl, 2)[lane] = vy;
#endif
  }
  wave_fence();
#if AEC_FILTERFAR_FIRST
  rdft_mixed_quad(wl, lane, T, 1 << 2);
  const float y_est = reinterpret_cast<float*>(tile(wl, 2))[64 + lane] * scale;
#else
  rdft_fwd_quad(wl, lane, T);
#endif
  unpack_tile(wl, 0, DFR, DFI, lane);
  unpack_tile(wl, 1, DWR, DWI, lane);
  wave_fence();

#if !AEC_FILTERFAR_FIRST
  xr[0] = fs_lane[0];
  xi[0] = fs_lane[1];
  AEC_FAR_SPECTRA()
  // ---- FilterFar (aec_core.c:147-169): partitions in order, per bin
  BINS_2TRIPS {
    float yr = 0.f, yi = 0.f;
#pragma unroll kFarUnroll
    for (int i = 0; i < kNumPart; ++i) {
      int px = i + op.xf_pos;
      if (px >= kNumPart) px -= kNumPart;
      float ar, ai, br, bi;
      if constexpr (kExtended) {
        ar = i == 0 ? XFR[bin] : ROW_LD((R_XF_RE + px));
        ai = i == 0 ? XFI[bin] : ROW_LD((R_XF_IM + px));
        br = ROW_LD((R_WF_RE + i));
        bi = ROW_LD((R_WF_IM + i));
      } else {
        ar = t_ == 0 ? xr[i] : (i == 0 ? XFR[64] : c64[R_XF_RE + px]);
        ai = t_ == 0 ? xi[i] : (i == 0 ? XFI[64] : c64[R_XF_IM + px]);
        br = t_ == 0 ? wr[i] : c64[R_WF_RE + i];
        bi = t_ == 0 ? wi[i] : c64[R_WF_IM + i];
      }
      yr += ar * br - ai * bi;
      yi += ar * bi + ai * br;
    }
    YFR[bin] = yr;
    YFI[bin] = yi;
  }
  wave_fence();


#endif
  AEC_STAMP(2)
  // ---- power smoothing and noise floor (aec_core.c:1144-1186)
  int noiseEstCtr = SCI(S_NOISEESTCTR);
  const bool noise_track = noiseEstCtr > 50;
  const bool noise_init = noiseEstCtr < 500 * mult;
  BINS_2TRIPS {
    const float xr = XFR[bin], xi = XFI[bin];
    const float far_spectrum = (xr * xr) + (xi * xi);
    const float near_spectrum = DFR[bin] * DFR[bin] + DFI[bin] * DFI[bin];
    const float xp = 0.9f * (t_ == 0 ? p_xpow : c64[R_XPOW]) + 0.1f * kNumPart * far_spectrum;
    const float dp = 0.9f * (t_ == 0 ? p_dpow : c64[R_DPOW]) + 0.1f * near_spectrum;
    if constexpr (BITS) {  // delay logging: spec_out = this block's two words of binary spectra
      if (spec_out != nullptr && t_ == 0)
        flow_binary_spectra(est, reinterpret_cast<unsigned*>(spec_out), far_spectrum, near_spectrum, EFR[lane], EFI[lane],
                            misc[6], misc[7], lane);
    } else if (spec_out != nullptr) {  // delay estimation on: |X|^2 and |D|^2 of the block for aec_delay_kernel (aec_core.c:1154-1155, 1191-1203)
      spec_out[bin] = far_spectrum;
      spec_out[kRow + bin] = near_spectrum;
    }
    ROW_ST(R_XPOW, xp);
    ROW_ST(R_DPOW, dp);
    XPW[bin] = xp;
    if constexpr (kMetrics) {  // UpdateLevel terms of farlevel / nearlevel (aec_core.c:612-618, 1270-1274)
      const bool edge = bin == 0 || bin == 64;
      MET[bin] = edge ? far_spectrum / 2 : far_spectrum;
      MET[kLRow + bin] = edge ? near_spectrum / 2 : near_spectrum;
    }
    float dmin = t_ == 0 ? p_dmin : c64[R_DMINPOW];
    if (noise_track) {
      if (dp < dmin) {
        dmin = (dp + 0.1f * (dmin - dp)) * 1.0002f;
      } else {
        dmin *= 1.0002f;
      }
      ROW_ST(R_DMINPOW, dmin);
    }
    float npow = dmin;
    if (noise_init) {
      float dinit = t_ == 0 ? p_dinit : c64[R_DINITMINPOW];
      if (dmin > dinit) {
        dinit = 0.999f * dinit + 0.001f * dmin;
      } else {
        dinit = dmin;
      }
      ROW_ST(R_DINITMINPOW, dinit);
      npow = dinit;
    }
    T1[bin] = npow;  // aec->noisePow for the comfort noise (kept until the NLP)
    // ---- buffer xf (aec_core.c:1203-1214)
    ROW_ST((R_XF_RE + op.xf_pos), xr);
    ROW_ST((R_XF_IM + op.xf_pos), xi);
  }
  if (noise_init) noiseEstCtr++;

  AEC_STAMP(3)
  // (mark 4 is taken at the call's entry, ahead of the far-end work: process_call)
  // ---- echo estimate and error (aec_core.c:1222-1238)
#if AEC_FILTERFAR_FIRST
  const float y = y_est;
#else
  {
    float2 v;
    v.x = YFR[lane];
    v.y = lane == 0 ? YFR[64] : YFI[lane];
    tile(wl, 0)[lane] = v;
  }
  wave_fence();
  rdft_inv_quad(wl, lane, T);
  const float y = reinterpret_cast<float*>(tile(wl, 0))[64 + lane] * scale;
#endif
  const float e = ne - y;
  wave_fence();

  AEC_STAMP(5)
  // ---- error fft (aec_core.c:1241-1254)
  ebuf[64 + lane] = e;
  {
    // zero-padded (aec_core.c:1241-1246) and windowed (aec_core.c:433-436) error spectra side by side
    float* tf = reinterpret_cast<float*>(tile(wl, 0));
    tf[lane] = 0.f;
    tf[64 + lane] = e;
    float* t1 = reinterpret_cast<float*>(tile(wl, 1));
    t1[lane] = ebuf[lane] * T.hann[lane];
    t1[64 + lane] = e * T.hann[64 - lane];
  }
  wave_fence();
  rdft_fwd_quad(wl, lane, T);
  unpack_tile(wl, 0, EFR, EFI, lane);
  unpack_tile(wl, 1, EWR, EWI, lane);
  wave_fence();

  AEC_STAMP(6)
  // ---- ScaleErrorSignal (aec_core.c:171-193)
  BINS_2TRIPS {
    float er = EFR[bin], ei = EFI[bin];
    if constexpr (kMetrics) {  // linoutlevel (aec_core.c:1258-1263)
      const float t = er * er + ei * ei;
      MET[2 * kLRow + bin] = (bin == 0 || bin == 64) ? t / 2 : t;
    }
    const float den = XPW[bin] + 1e-10f;
    er = fdiv_aec(er, den);
    ei = fdiv_aec(ei, den);
    float abs_ef = sqrtf(er * er + ei * ei);
    if (abs_ef > error_threshold) {
      abs_ef = fdiv_aec(error_threshold, abs_ef + 1e-10f);
      er *= abs_ef;
      ei *= abs_ef;
    }
    er *= mu;
    ei *= mu;
    EFR[bin] = er;
    EFI[bin] = ei;
  }
  wave_fence();

  AEC_STAMP(7)
  // the PSD rows of the NLP: in flight during the six FFT rounds of the filter update
  const float q_sd = sld(sb, ROWO(R_SD), lane), q_se = sld(sb, ROWO(R_SE), lane);
  const float q_sx = sld(sb, ROWO(R_SX), lane);
  const float q_sde_r = sld(sb, ROWO(R_SDE_RE), lane), q_sde_i = sld(sb, ROWO(R_SDE_IM), lane);
  const float q_sxd_r = sld(sb, ROWO(R_SXD_RE), lane), q_sxd_i = sld(sb, ROWO(R_SXD_IM), lane);
  // ---- FilterAdaptation (aec_core.c:221-269): four partitions per round, streamed: a group's far
  // spectra and filter rows are requested one group ahead (partition 0 = this block's spectrum, still in
  // registers), the updated filter rows go straight back
  float gx[2][8], gw[2][8];  // [buffer][re0 im0 re1 im1 ...] of a group's four partitions
#define AEC_LOAD_GROUP(buf, g_)                                                     \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                   \
    const int i_ = 4 * (g_) + k;                                                    \
    int px_ = i_ + op.xf_pos;                                                       \
    if (px_ >= kNumPart) px_ -= kNumPart;                                           \
    gx[buf][2 * k] = i_ == 0 ? fs_lane[0] : sld(sb, ROWO((R_XF_RE + px_)), lane);   \
    gx[buf][2 * k + 1] = i_ == 0 ? fs_lane[1] : sld(sb, ROWO((R_XF_IM + px_)), lane); \
    gw[buf][2 * k] = sld(sb, ROWO((R_WF_RE + i_)), lane);                           \
    gw[buf][2 * k + 1] = sld(sb, ROWO((R_WF_IM + i_)), lane);                       \
  }
  AEC_LOAD_GROUP(0, 0)
  // The next block's FilterFar (aec_core.c:147-169) rides along: its partition i multiplies the far spectrum this
  // block holds as partition i - 1 (the history moves by one block; partition 0 = the next block's own spectrum)
  // with the filter partition i as this update leaves it -- both in registers here, in the reference's partition
  // order.  The sums wait in the YFR / YFI rows, and the next block skips its 46 row loads.
  const bool carry_out = next_slot != nullptr;
  float cy_r = 0.f, cy_i = 0.f, nx_r = 0.f, nx_i = 0.f;
  if (carry_out) {
    if constexpr (FLOW) {
      const StateBufT<kSc1> nb = state_buf<kSc1>(next_slot, kFarSlotDwords);
      nx_r = sld(nb, 0, lane);
      nx_i = sld(nb, kRow, lane);
    } else {
      nx_r = next_slot[lane];
      nx_i = next_slot[kRow + lane];
    }
  }
#pragma unroll
  for (int g = 0; g < kNumPart / 4; ++g) {
    const int cb = g & 1;
    if (g + 1 < kNumPart / 4) {
      AEC_LOAD_GROUP((cb ^ 1), (g + 1))
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = 4 * g + k;
      int px = i + op.xf_pos;
      if (px >= kNumPart) px -= kNumPart;
      const float ar = gx[cb][2 * k], ai = -gx[cb][2 * k + 1];
      float2 v;
      v.x = ar * EFR[lane] - ai * EFI[lane];
      v.y = ar * EFI[lane] + ai * EFR[lane];
      {
        const float cr = i == 0 ? XFR[64] : c64[R_XF_RE + px];
        const float ci = -(i == 0 ? XFI[64] : c64[R_XF_IM + px]);
        const float p64 = cr * EFR[64] - ci * EFI[64];
        v.y = lane == 0 ? p64 : v.y;
      }
      tile(wl, k)[lane] = v;
    }
    wave_fence();
    rdft_inv_quad<true>(wl, lane, T);  // scaled first half, zeroed second half (aec_core.c:245-254)
    rdft_fwd_quad(wl, lane, T);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = 4 * g + k;
      const float2 v = tile(wl, k)[lane];
      // lane 0 carries (fft[0], fft[1]) = updates of the real parts of bins 0 and 64
      const float wr_new = gw[cb][2 * k] + v.x;
      const float wi_sum = gw[cb][2 * k + 1] + v.y;
      const float wi_new = lane == 0 ? gw[cb][2 * k + 1] : wi_sum;
      if (lane == 0) c64[R_WF_RE + i] += v.y;
      sst(sb, ROWO((R_WF_RE + i)), lane, wr_new);
      sst(sb, ROWO((R_WF_IM + i)), lane, wi_new);
      if (carry_out) {
        cy_r += nx_r * wr_new - nx_i * wi_new;
        cy_i += nx_r * wi_new + nx_i * wr_new;
      }
      nx_r = gx[cb][2 * k];
      nx_i = gx[cb][2 * k + 1];
    }
    wave_fence();
  }
#undef AEC_LOAD_GROUP
  if (carry_out) {
    float nx64_r, nx64_i;
    if constexpr (FLOW) {
      const StateBufT<kSc1> nb = state_buf<kSc1>(next_slot, kFarSlotDwords);
      nx64_r = sld(nb, 64, 0);
      nx64_i = sld(nb, kRow + 64, 0);
    } else {
      nx64_r = next_slot[64];
      nx64_i = next_slot[kRow + 64];
    }
    float yr = 0.f, yi = 0.f;  // bin 64: the column copy holds this block's spectrum and the updated filter
#pragma unroll kFarUnroll
    for (int i = 0; i < kNumPart; ++i) {
      int px = i + next_xf_pos;
      if (px >= kNumPart) px -= kNumPart;
      const float ar = i == 0 ? nx64_r : c64[R_XF_RE + px], ai = i == 0 ? nx64_i : c64[R_XF_IM + px];
      const float br = c64[R_WF_RE + i], bi = c64[R_WF_IM + i];
      yr += ar * br - ai * bi;
      yi += ar * bi + ai * br;
    }
    YFR[lane] = cy_r;
    YFI[lane] = cy_i;
    YFR[64] = yr;
    YFI[64] = yi;
    wave_fence();
  }

  AEC_STAMP(8)
  // =================================================== NonLinearProcessing (aec_core.c:852-1082)
  int delayEstCtr = SCI(S_DELAYESTCTR) + 1;
  if (delayEstCtr == 10 * mult) delayEstCtr = 0;
  int delayIdx = SCI(S_DELAYIDX);
  if (delayEstCtr == 0) {  // PartitionDelay (aec_core.c:294-318)
    float* pe = wl + kLdsTile;  // 12 rows of 66 overlaying the tiles and the dead rows
    float* pen = kExtended ? c64 + kPeScratch : misc;  // the partitions' energies
    for (int c0 = 0; c0 < kNumPart; c0 += kPeChunk) {
      const int nc = kNumPart - c0 < kPeChunk ? kNumPart - c0 : kPeChunk;
      for (int i = 0; i < nc; ++i)
        BINS_2TRIPS {
          const float wr = ROW_LD((R_WF_RE + c0 + i)), wi = ROW_LD((R_WF_IM + c0 + i));
          pe[i * kLRow + bin] = wr * wr + wi * wi;
        }
      wave_fence();
      if (lane < nc) {
        float wfEn = 0.f;
#pragma unroll
        for (int j = 0; j < 65; ++j) wfEn += pe[lane * kLRow + j];
        pen[c0 + lane] = wfEn;
      }
      wave_fence();
    }
    float wfEnMax = 0.f;
    delayIdx = 0;
    for (int i = 0; i < kNumPart; ++i) {
      const float v = pen[i];
      if (v > wfEnMax) {
        wfEnMax = v;
        delayIdx = i;
      }
    }
    wave_fence();
  }

  AEC_STAMP(9)
  // (the windowed near / error spectra were computed next to the plain ones above)

  AEC_STAMP(10)
  // ---- SmoothedPSD + coherence (aec_core.c:332-385, 438-448)
  // aec_core.c:111-114, 337-339: WebRtcAec_kNormal / kExtendedSmoothingCoefficients[mult - 1]
  const float g0 = mult == 1 ? 0.9f : kExtended ? 0.92f : 0.93f, g1 = mult == 1 ? 0.1f : kExtended ? 0.08f : 0.07f;
  int pd = op.xfw_head + delayIdx;  // the history ring is 32 deep whatever the filter length
  if (pd >= kNumPartMax) pd -= kNumPartMax;
  BINS_2TRIPS {
    const float dr = DWR[bin], di = DWI[bin], er = EWR[bin], ei = EWI[bin];
    const float xr = delayIdx == 0 ? XWR[bin] : ROW_LD((R_XFW + 2 * pd));
    const float xi = delayIdx == 0 ? XWI[bin] : ROW_LD((R_XFW + 2 * pd + 1));
    const float sd = g0 * (t_ == 0 ? q_sd : c64[R_SD]) + g1 * (dr * dr + di * di);
    const float se = g0 * (t_ == 0 ? q_se : c64[R_SE]) + g1 * (er * er + ei * ei);
    const float xx = xr * xr + xi * xi;
    const float sx = g0 * (t_ == 0 ? q_sx : c64[R_SX]) + g1 * (xx > 15.f ? xx : 15.f);
    const float sde_r = g0 * (t_ == 0 ? q_sde_r : c64[R_SDE_RE]) + g1 * (dr * er + di * ei);
    const float sde_i = g0 * (t_ == 0 ? q_sde_i : c64[R_SDE_IM]) + g1 * (dr * ei - di * er);
    const float sxd_r = g0 * (t_ == 0 ? q_sxd_r : c64[R_SXD_RE]) + g1 * (dr * xr + di * xi);
    const float sxd_i = g0 * (t_ == 0 ? q_sxd_i : c64[R_SXD_IM]) + g1 * (dr * xi - di * xr);
    ROW_ST(R_SD, sd);
    ROW_ST(R_SE, se);
    ROW_ST(R_SX, sx);
    ROW_ST(R_SDE_RE, sde_r);
    ROW_ST(R_SDE_IM, sde_i);
    ROW_ST(R_SXD_RE, sxd_r);
    ROW_ST(R_SXD_IM, sxd_i);
    T0[bin] = sd;
    XPW[bin] = se;
    COHDE[bin] = fdiv_aec(sde_r * sde_r + sde_i * sde_i, sd * se + 1e-10f);
    COHXD[bin] = fdiv_aec(sxd_r * sxd_r + sxd_i * sxd_i, sx * sd + 1e-10f);
  }
  wave_fence();
  const int prefBandSize = 24 / mult, minPrefBand = 4 / mult;
  if (lane < 4) {
    // the four sequential sums of the reference, one lane each: lanes 0, 1 all 65 bins of sd / se, lanes 2, 3 the
    // preferred band of cohxd / cohde, read from its first bin on.  The band's length (24 / mult terms) is the part all
    // four lanes share; the rest is lanes 0, 1 alone -- straight runs of adds, no per-term range test
    const float* src = lane == 0 ? T0 : lane == 1 ? XPW : (lane == 2 ? COHXD : COHDE) + minPrefBand;
    float acc = 0.f;
    if (mult == 1) {
#pragma unroll
      for (int j = 0; j < 24; ++j) acc += src[j];
      if (lane < 2) {
#pragma unroll
        for (int j = 24; j < 65; ++j) acc += src[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) acc += src[j];
      if (lane < 2) {
#pragma unroll
        for (int j = 12; j < 65; ++j) acc += src[j];
      }
    }
    misc[lane] = acc;
  }
  wave_fence();
  const float sdSum = misc[0], seSum = misc[1];
  float hNlXdAvg = misc[2], hNlDeAvg = misc[3];
  wave_fence();
  int divergeState = SCI(S_DIVERGESTATE);
  divergeState = (divergeState ? 1.05f : 1.0f) * seSum > sdSum;
  if (divergeState) {
    BINS_2TRIPS {
      EWR[bin] = DWR[bin];
      EWI[bin] = DWI[bin];
    }
  }
  if (!kExtended && seSum > (19.95f * sdSum)) {  // aec_core.c:383
    for (int i = 0; i < 2 * kNumPart; ++i)
      BINS_2TRIPS ROW_ST((R_WF_RE + i), 0.f);
  }

  AEC_STAMP(11)
  // ---- aec_core.c:895-960
  hNlXdAvg /= prefBandSize;
  hNlXdAvg = 1 - hNlXdAvg;
  hNlDeAvg /= prefBandSize;
  float hNlXdAvgMin = SCF(S_HNLXDAVGMIN);
  int stNearState = SCI(S_STNEARSTATE);
  int echoState;
  float overDrive = SCF(S_OVERDRIVE);
  if (hNlXdAvg < 0.75f && hNlXdAvg < hNlXdAvgMin) hNlXdAvgMin = hNlXdAvg;
  if (hNlDeAvg > 0.98f && hNlXdAvg > 0.9f) {
    stNearState = 1;
  } else if (hNlDeAvg < 0.95f || hNlXdAvg < 0.8f) {
    stNearState = 0;
  }
  // kNormalMinOverDrive / kExtendedMinOverDrive (aec_core.c:108-109, 872-874)
  const float minOverDrive = kExtended ? (nlp_mode == 0 ? 3.0f : nlp_mode == 1 ? 6.0f : 15.0f)
                                       : (nlp_mode == 0 ? 1.0f : nlp_mode == 1 ? 2.0f : 5.0f);
  const float targetSupp = nlp_mode == 0 ? -6.9f : nlp_mode == 1 ? -11.5f : -18.4f;
  float hNlFb, hNlFbLow;
  int hnl_kind;  // 0: cohde, 1: 1 - cohxd, 2: min of both
  if (hNlXdAvgMin == 1) {
    echoState = 0;
    overDrive = minOverDrive;
    if (stNearState == 1) {
      hnl_kind = 0;
      hNlFb = hNlDeAvg;
      hNlFbLow = hNlDeAvg;
    } else {
      hnl_kind = 1;
      hNlFb = hNlXdAvg;
      hNlFbLow = hNlXdAvg;
    }
  } else if (stNearState == 1) {
    echoState = 0;
    hnl_kind = 0;
    hNlFb = hNlDeAvg;
    hNlFbLow = hNlDeAvg;
  } else {
    echoState = 1;
    hnl_kind = 2;
    hNlFb = 0.f;
    hNlFbLow = 0.f;
  }
  BINS_2TRIPS {
    const float a = COHDE[bin], b = 1 - COHXD[bin];
    HNL[bin] = hnl_kind == 0 ? a : hnl_kind == 1 ? b : (a < b ? a : b);
  }
  wave_fence();
  if (hnl_kind == 2) {
    // order statistics of the preferred band (qsort + pick, aec_core.c:953-959) by rank counting
    const int iFb = (int)floorf(0.75f * (prefBandSize - 1)), iLow = (int)floorf(0.5f * (prefBandSize - 1));
    if (lane < prefBandSize) {
      const float v = HNL[minPrefBand + lane];
      int rank = 0;
      // a laundered copy of the lane index: the 24 `j < lane` masks are block-invariant, and hoisted out of
      // the block loop they took 48 SGPRs (which spilled into VGPR lanes, which spilled to scratch)
      int ln = lane, pbs = prefBandSize;  // pbs: the band size as a vector value, for the same reason
      asm volatile("" : "+v"(ln), "+v"(pbs));
#pragma unroll
      for (int j = 0; j < 24; ++j) {
        const float u = HNL[minPrefBand + j];  // j < 24 + 4 stays inside the row
        rank += (int)(j < pbs) & ((int)(u < v) | ((int)(u == v) & (int)(j < ln)));  // no short-circuit branches
      }
      if (rank == iFb) misc[4] = v;
      if (rank == iLow) misc[5] = v;
    }
    wave_fence();
    hNlFb = misc[4];
    hNlFbLow = misc[5];
    wave_fence();
  }

  // ---- aec_core.c:962-992
  float hNlFbLocalMin = SCF(S_HNLFBLOCALMIN), hNlFbMin = SCF(S_HNLFBMIN);
  int hNlNewMin = SCI(S_HNLNEWMIN), hNlMinCtr = SCI(S_HNLMINCTR);
  float overDriveSm = SCF(S_OVERDRIVESM);
  if (hNlFbLow < 0.6f && hNlFbLow < hNlFbLocalMin) {
    hNlFbLocalMin = hNlFbLow;
    hNlFbMin = hNlFbLow;
    hNlNewMin = 1;
    hNlMinCtr = 0;
  }
  {
    const float a = hNlFbLocalMin + 0.0008f / mult, b = hNlXdAvgMin + 0.0006f / mult;
    hNlFbLocalMin = a < 1 ? a : 1;
    hNlXdAvgMin = b < 1 ? b : 1;
  }
  if (hNlNewMin == 1) hNlMinCtr++;
  if (hNlMinCtr == 2) {
    hNlNewMin = 0;
    hNlMinCtr = 0;
    const float v = targetSupp / ((float)log((double)(hNlFbMin + 1e-10f)) + 1e-10f);
    overDrive = v > minOverDrive ? v : minOverDrive;
  }
  if (overDrive < overDriveSm) {
    overDriveSm = 0.99f * overDriveSm + 0.01f * overDrive;
  } else {
    overDriveSm = 0.9f * overDriveSm + 0.1f * overDrive;
  }

  AEC_STAMP(12)
  // ---- OverdriveAndSuppress + ComfortNoise (aec_core.c:271-292, 461-500)
  const uint32_t seed = (uint32_t)SCI(S_SEED);
  // comfort-noise phase of every bin but bin 0 (which has none): lane q > 0 evaluates bin q, lane 0 evaluates
  // bin 64, so that the second trip (bin 64 on every lane) needs no sincos of its own
  float2 sc_lane;
  {
    const int nb = lane == 0 ? 64 : lane;
    // rand[nb - 1]: the LCG after `nb` steps (randomization_functions.c:93-115)
    const uint32_t s = (T.lcg_a[nb - 1] * seed + T.lcg_c[nb - 1]) & 0x7fffffffu;
    const float rnd = ((float)(int16_t)(s >> 16)) / 32768;
    sc_lane = nlp_sincos(6.28318530717959f * rnd);
  }
  BINS_2TRIPS {
    float h = HNL[bin];
    if (h > hNlFb) h = T.weight[bin] * hNlFb + (1 - T.weight[bin]) * h;
    h = nlp_pow(h, overDriveSm * T.odrive[bin], exp2_global);
    float er = EWR[bin] * h, ei = EWI[bin] * h;
    ei *= -1;
    float ur = 0.f, ui = 0.f;
    if (bin > 0) {
      const float noise = sqrtf(T1[bin]);
      float2 sc2 = sc_lane;
      if (t_ == 1) {  // bin 64: lane 0's evaluation
        sc2.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc_lane.x), 0));
        sc2.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc_lane.y), 0));
      }
      ur = noise * sc2.y;
      ui = -noise * sc2.x;
      if (bin == 64) ui = 0.f;
      if (num_high > 0) {  // kept for the high band's comfort noise (aec_core.c:501-545)
        DFR[bin] = noise;
        YFR[bin] = sc2.y;
        YFI[bin] = sc2.x;
      }
    }
    const float r = 1 - h * h;
    const float tmp2 = sqrtf(r > 0 ? r : 0);
    er += tmp2 * ur;
    ei += tmp2 * ui;
    EWR[bin] = er;
    EWI[bin] = ei;
    if constexpr (kMetrics) {  // nlpoutlevel (aec_core.c:1000-1006)
      const float t = er * er + ei * ei;
      MET[3 * kLRow + bin] = (bin == 0 || bin == 64) ? t / 2 : t;
    }
    if (num_high > 0) {
      HNL[bin] = h;
      DFI[bin] = tmp2;
    }
  }
  const uint32_t new_seed = (T.lcg_a[63] * seed + T.lcg_c[63]) & 0x7fffffffu;
  wave_fence();

  AEC_STAMP(13)
  // ---- inverse error fft, overlap-add, saturation (aec_core.c:1006-1030)
  {
    float2 v;
    v.x = EWR[lane];
    v.y = lane == 0 ? EWR[64] : -EWI[lane];
    tile(wl, 0)[lane] = v;
  }
  wave_fence();
  rdft_inv_quad(wl, lane, T);
  {
    const float* tf = reinterpret_cast<const float*>(tile(wl, 0));
    float a = tf[lane] * scale;
    a = a * T.hann[lane] + p_outbuf;
    const float b = tf[64 + lane] * scale;
    sst(sb, kOffOutBuf, lane, b * T.hann[64 - lane]);
    const float o = a > 32767.f ? 32767.f : (a < -32768.f ? -32768.f : a);
    sst(sb, kOffOutFr, ring_idx(op.out_wpos, lane, kFrBufLen), o);
  }
  if (num_high > 0) high_band_block(st, wl, T, op.near_rpos, op.out_wpos, lane);
  if constexpr (kMetrics) metrics_block(met, MET, lane, echoState);
  AEC_STAMP(14)
  // ---- carry the block (aec_core.c:1069-1081; the xfwBuf shift is the host's circular head)
  sst(sb, kOffDBuf, lane, ne);
  sst(sb, kOffEBuf, lane, e);
  {
    SC_SETF(S_HNLFBMIN, hNlFbMin);
    SC_SETF(S_HNLFBLOCALMIN, hNlFbLocalMin);
    SC_SETF(S_HNLXDAVGMIN, hNlXdAvgMin);
    SC_SETF(S_OVERDRIVE, overDrive);
    SC_SETF(S_OVERDRIVESM, overDriveSm);
    SC_SETI(S_HNLNEWMIN, hNlNewMin);
    SC_SETI(S_HNLMINCTR, hNlMinCtr);
    SC_SETI(S_DELAYIDX, delayIdx);
    SC_SETI(S_STNEARSTATE, stNearState);
    SC_SETI(S_ECHOSTATE, echoState);
    SC_SETI(S_DIVERGESTATE, divergeState);
    SC_SETI(S_NOISEESTCTR, noiseEstCtr);
    SC_SETI(S_DELAYESTCTR, delayEstCtr);
    SC_SETI(S_SEED, (int)new_seed);
    if (lane < 32) sst(sb, kOffScalars, lane, scrow);
  }
  wave_fence();
#pragma unroll
  for (int k = 0; k < kC64Chunks; ++k)
    if (64 * (k + 1) <= kC64Lds || 64 * k + lane < kC64Lds) sst(sb, kOffC64 + 64 * k, lane, c64[64 * k + lane]);
  AEC_STAMP(15)
#undef AEC_STAMP
#undef AEC_LOAD_POWER_ROWS
#undef AEC_FAR_SPECTRA
#undef ROW_LD
#undef ROWO
#undef ROW_ST
#undef SCI
#undef SCF
#undef SC_SETI
#undef SC_SETF
}

// One WebRtcAec_Process call of one stream (running phase), after the far-end work that precedes it: per
// 80-sample sub-frame append the near samples, run the scheduled blocks, emit 80 output samples.  OPS / FOPS:
// the call's descriptors, in the kernel's arguments (plain build) or in device memory (hand-off build).
// AGN: the delay-agnostic mode inside the call (agn != nullptr): per sub-frame the stream's own far-buffer control
// step, which picks the far slots of its blocks, then the blocks, then the estimator's share of them.
template <bool kMetrics, int NP, bool FLOW, bool AGN = false, class OPS, class FOPS, class AGNP = const AgnOps*>
__device__ __forceinline__ void process_call(float* __restrict__ st, float* far_ring, float* __restrict__ wl,
                                             const SharedTables& T, const AecTables* __restrict__ G,
                                             const float* __restrict__ nin, float* __restrict__ o, int num_streams,
                                             int nrOfSamples, int stream, int lane, const OPS& ops,
                                             const float* __restrict__ farend, const FOPS& fops,
                                             const float* near_high, float* out_high, float* met,
                                             unsigned long long* __restrict__ stamps, float* spectra,
                                             const DelayBlock* dblocks, AGNP agn = nullptr) {
  constexpr int kAux = FLOW ? kSc1 : 0;
  constexpr int kDwords = AecRows(NP).state_dwords;
  [[maybe_unused]] const StateBufT<kAux> sb = state_buf<kAux>(st, kDwords);
  if (stamps != nullptr && stream == 0) stamps[4] = __builtin_amdgcn_s_memtime();  // diagnostic: the call's entry
  if (farend != nullptr) {
    // the WebRtcAec_BufferFarend call that preceded this Process call, fused into the launch
    farend_work<FLOW>(st, far_ring, wl, T, farend, num_streams, stream, fops, lane, kDwords);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  int blk = 0;  // blocks of this launch so far
  bool carried = false;  // the coming block finds its echo estimate's spectrum in LDS
  // (not with the high band, whose comfort noise borrows the YFR / YFI rows; not in the delay-agnostic mode, where
  // the next block's far slot is the estimator's to choose)
  [[maybe_unused]] const bool carry_ok = !AGN && (FLOW || (ops.num_high == 0 && !ops.agnostic));
  for (int s = 0; s < ops.nsub; ++s) {
    const auto& sf = ops.sub[s];
    // near samples of this sub-frame are read into registers first: `out` may alias `nearend`
    const float n0 = nin[80 * s + lane];
    const float n1 = lane < 16 ? nin[80 * s + 64 + lane] : 0.f;
    if constexpr (FLOW) {
      sst(sb, kOffNearFr, ring_idx(sf.near_wpos, lane, kFrBufLen), n0);
      if (lane < 16) sst(sb, kOffNearFr, ring_idx(sf.near_wpos, 64 + lane, kFrBufLen), n1);
    } else {
      st[kOffNearFr + ring_idx(sf.near_wpos, lane, kFrBufLen)] = n0;
      if (lane < 16) st[kOffNearFr + ring_idx(sf.near_wpos, 64 + lane, kFrBufLen)] = n1;
    }
    if (!FLOW && ops.num_high > 0) {  // the high band's frame into its ring (aec_core.c:1691-1693)
      const float* hin = near_high + (size_t)stream * nrOfSamples;
      const float h0 = hin[80 * s + lane];
      const float h1 = lane < 16 ? hin[80 * s + 64 + lane] : 0.f;
      st[kOffNearFrH + ring_idx(sf.near_wpos, lane, kFrBufLen)] = h0;
      if (lane < 16) st[kOffNearFrH + ring_idx(sf.near_wpos, 64 + lane, kFrBufLen)] = h1;
    }
    // ring traffic between lanes of this wave goes through L2: order it at workgroup scope (same CU and L1; agent scope would flush the XCD L2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    [[maybe_unused]] int agn_slot0 = 0, agn_slot1 = 0;
    [[maybe_unused]] float* agn_bits = wl + kLdsMisc + 12;  // the sub-frame's binary spectra: 2 x 2 words behind the NLP's scratch
    if constexpr (AGN) {
      DelayBlock* eb = const_cast<DelayBlock*>(dblocks) + stream;
      aspaec_est::Hist H;
      aspaec_est::Scalars sc;
      aspaec_est::load_estimator<FLOW>(&eb->s, H, sc, lane);
      aspaec_est::control_step<FLOW>(eb, H, sc, agn->sub[s], lane, agn_slot0, agn_slot1);
      aspaec_est::store_estimator<false, FLOW>(&eb->s, H, sc, lane);
    }
    for (int k = 0; k < sf.nblocks; ++k) {
      const auto& op = sf.blk[k];
      // the block after this one, when this call holds one: its echo estimate is accumulated during this block's
      // filter update (process_block, carry_out)
      const float* next_slot = nullptr;
      int next_xf = 0;
#if AEC_CARRY
      if (carry_ok) {
        int nslot = -1;
        if (k + 1 < sf.nblocks) {
          nslot = sf.blk[k + 1].far_slot;
          next_xf = sf.blk[k + 1].xf_pos;
        } else if (s + 1 < ops.nsub && ops.sub[s + 1].nblocks > 0) {
          nslot = ops.sub[s + 1].blk[0].far_slot;
          next_xf = ops.sub[s + 1].blk[0].xf_pos;
        }
        if (nslot >= 0 && next_xf == (op.xf_pos == 0 ? NP - 1 : op.xf_pos - 1))
          next_slot = far_ring + ((size_t)nslot * num_streams + stream) * kFarSlotDwords;
      }
#endif
      // the far slot: the batch's (lock-step), or in the delay-agnostic mode the stream's own (aec_delay_kernel)
      const int far_slot = AGN ? (k == 0 ? agn_slot0 : agn_slot1)
                               : (!FLOW && ops.agnostic) ? __builtin_amdgcn_readfirstlane(dblocks[stream].slot[blk & 1]) : op.far_slot;
      const float* slot = far_ring + ((size_t)far_slot * num_streams + stream) * kFarSlotDwords;
      float* spec_out = AGN ? (agn->logging ? agn_bits + 2 * k : nullptr)
                        : FLOW ? (spectra != nullptr ? spectra + 2 * blk : nullptr)  // hand-off build: `spectra` = this step's binary-spectra words
                             : ops.spectra ? spectra + ((size_t)stream * kSpecBlocks + (blk & (kSpecBlocks - 1))) * kSpecDwords : nullptr;
      ++blk;
      process_block<kMetrics, NP, FLOW, FLOW || AGN>(st, wl, slot, T, op, ops.mult, ops.nlp_mode, ops.mu, ops.error_threshold, lane, G->exp2_64,
                    FLOW ? 0 : ops.num_high, met,
                    (stamps != nullptr && stream == 0 && s == 0 && k == 0) ? stamps : nullptr,  // wave-uniform; every lane stores the same scalar time
                    spec_out, carried, next_slot, next_xf, (FLOW || AGN) && dblocks != nullptr ? dblocks + stream : nullptr);
      carried = next_slot != nullptr;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if constexpr (AGN) {
      if (agn->logging && sf.nblocks > 0) {  // the estimator's share of this sub-frame's blocks (aec_core.c:1191-1203)
        DelayBlock* eb = const_cast<DelayBlock*>(dblocks) + stream;
        aspaec_est::Hist H;
        aspaec_est::Scalars sc;
        aspaec_est::load_estimator<FLOW>(&eb->s, H, sc, lane);
        for (int k = 0; k < sf.nblocks; ++k) {
          const unsigned bfar = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(agn_bits[2 * k]));
          const unsigned bnear = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(agn_bits[2 * k + 1]));
          const int delay_estimate = aspaec_est::estimator_block_bits(H, sc, bfar, bnear, lane);
          if (delay_estimate >= 0 && lane == 0) {
            int32_t* cnt = &eb->s.delay_histogram[delay_estimate];
            aspaec_est::est_st<FLOW>(cnt, aspaec_est::est_ld<FLOW>(cnt) + 1);
          }
        }
        aspaec_est::store_estimator<false, FLOW>(&eb->s, H, sc, lane);
      }
    }
    if constexpr (FLOW) {
      o[80 * s + lane] = sld(sb, kOffOutFr, ring_idx(sf.out_rpos, lane, kFrBufLen));
      if (lane < 16) o[80 * s + 64 + lane] = sld(sb, kOffOutFr, ring_idx(sf.out_rpos, 64 + lane, kFrBufLen));
    } else {
      o[80 * s + lane] = st[kOffOutFr + ring_idx(sf.out_rpos, lane, kFrBufLen)];
      if (lane < 16) o[80 * s + 64 + lane] = st[kOffOutFr + ring_idx(sf.out_rpos, 64 + lane, kFrBufLen)];
    }
    if (!FLOW && ops.num_high > 0) {  // aec_core.c:1774-1776
      float* ho = out_high + (size_t)stream * nrOfSamples;
      ho[80 * s + lane] = st[kOffOutFrH + ring_idx(sf.out_rpos, lane, kFrBufLen)];
      if (lane < 16) ho[80 * s + 64 + lane] = st[kOffOutFrH + ring_idx(sf.out_rpos, 64 + lane, kFrBufLen)];
    }
  }
}

// WebRtcAec_ProcessFrames for every stream (running phase).
#ifndef AEC_WAVES
#define AEC_WAVES 4  // waves per SIMD the register allocation aims at (4: every stream of a 4096-stream batch is resident at once)
#endif
template <bool kMetrics, int NP>
__global__ __launch_bounds__(256, kMetrics ? 2 : NP == kNumPartNormal ? AEC_WAVES : 3) void aec_process_kernel(int stream0, int stream_end, float* __restrict__ state,
                                                          float* far_ring,
                                                          const AecTables* __restrict__ G,
                                                          const float* __restrict__ nearend,
                                                          float* __restrict__ out, int num_streams,
                                                          int nrOfSamples, ProcOps ops,
                                                          const float* __restrict__ farend, FarOps fops,
                                                          const float* near_high, float* out_high,
                                                          float* metrics,
                                                          unsigned long long* __restrict__ stamps,
                                                          float* spectra, const DelayBlock* dblocks) {
  __shared__ SharedTables T;
  constexpr int kWaveLds = lds_wave(NP, kMetrics);
  __shared__ float lds[4 * kWaveLds];
  stage_tables(T, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: the stream's state block and LDS region get scalar bases
  const int stream = stream0 + blockIdx.x * 4 + wave;  // this launch covers streams stream0 .. stream_end - 1
  if (stream >= stream_end) return;
  float* wl = lds + wave * kWaveLds;
  float* st = state + (size_t)stream * AecRows(NP).state_dwords;
  const float* nin = nearend + (size_t)stream * nrOfSamples;
  float* o = out + (size_t)stream * nrOfSamples;
  float* met = kMetrics ? metrics + (size_t)stream * kMetDwords : nullptr;
  process_call<kMetrics, NP, false>(st, far_ring, wl, T, G, nin, o, num_streams, nrOfSamples, stream, lane, ops, farend, fops,
                                    near_high, out_high, met, stamps, spectra, dblocks);
}

// The delay-agnostic mode in one launch per call (one band, no metrics): the control steps and the estimator run in
// the stream's own wave (process_call<AGN>), so a call is not cut into a launch per sub-frame with estimator / control
// launches between them.
template <int NP>
__global__ __launch_bounds__(256, NP == kNumPartNormal ? AEC_WAVES : 3) void aec_process_agn_kernel(
    int stream0, int stream_end, float* __restrict__ state, float* far_ring, const AecTables* __restrict__ G,
    const float* __restrict__ nearend, float* __restrict__ out, int num_streams, int nrOfSamples, ProcOps ops,
    const float* __restrict__ farend, FarOps fops, DelayBlock* dblocks, AgnOps agn) {
  __shared__ SharedTables T;
  constexpr int kWaveLds = lds_wave(NP, false);
  __shared__ float lds[4 * kWaveLds];
  stage_tables(T, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = stream0 + blockIdx.x * 4 + wave;
  if (stream >= stream_end) return;
  float* wl = lds + wave * kWaveLds;
  float* st = state + (size_t)stream * AecRows(NP).state_dwords;
  const float* nin = nearend + (size_t)stream * nrOfSamples;
  float* o = out + (size_t)stream * nrOfSamples;
  process_call<false, NP, false, true>(st, far_ring, wl, T, G, nin, o, num_streams, nrOfSamples, stream, lane, ops,
                                                        farend, fops, nullptr, nullptr, nullptr, nullptr, nullptr, dblocks, &agn);
}

// The hand-off build: grid (groups of four streams, frame steps); see AecFlowArgs.  One band, no metrics, no
// delay estimation (aec_api.hip keeps the other configurations on the plain build).
template <int NP, bool AGN = false>
__global__ __launch_bounds__(256, NP == kNumPartNormal ? AEC_WAVES : 3) void aec_process_flow_kernel(
    float* __restrict__ state, float* far_ring, const AecTables* __restrict__ G, int num_streams, int nrOfSamples,
    AecFlowArgs fa) {
  __shared__ SharedTables T;
  constexpr int kWaveLds = lds_wave(NP, false);
  __shared__ float lds[4 * kWaveLds];
  stage_tables(T, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wave;
  if (stream >= num_streams) return;
  // the step's descriptor: read-only for the whole launch, so it may come through the scalar cache
  typedef const __attribute__((address_space(4))) AecFlowStep* StepPtr;
  typedef const __attribute__((address_space(4))) AecFlowStepAgn* StepAgnPtr;
  // (delay-agnostic mode: the array holds AecFlowStepAgn elements, the step first)
  const StepPtr fs = AGN ? (StepPtr)(reinterpret_cast<const AecFlowStepAgn*>(fa.steps) + blockIdx.y) : (StepPtr)(fa.steps + blockIdx.y);
  const unsigned want = fa.want + blockIdx.y;
  float* wl = lds + wave * kWaveLds;
  float* st = state + (size_t)stream * AecRows(NP).state_dwords;
  const float* nin = fs->nearend + (size_t)stream * nrOfSamples;
  float* o = fs->out + (size_t)stream * nrOfSamples;
  if (!flow_wait(fa, want, stream, lane)) return;
  const int spec_base = fs->spec_base;
  float* bits = (fa.bits != nullptr && spec_base >= 0)
                    ? reinterpret_cast<float*>(fa.bits + ((size_t)stream * kFlowBitsBlocks + spec_base) * 2) : nullptr;
  if constexpr (AGN) {
    typedef const __attribute__((address_space(4))) AgnOps* AgnPtr;
    const AgnPtr agn = &((StepAgnPtr)(reinterpret_cast<const AecFlowStepAgn*>(fa.steps) + blockIdx.y))->agn;
    process_call<false, NP, true, true>(
        st, far_ring, wl, T, G, nin, o, num_streams, nrOfSamples, stream, lane, fs->ops, fs->farend, fs->fops, nullptr, nullptr,
        nullptr, nullptr, nullptr, fa.est, agn);
  } else {
    process_call<false, NP, true>(st, far_ring, wl, T, G, nin, o, num_streams, nrOfSamples, stream, lane, fs->ops,
                                  fs->farend, fs->fops, nullptr, nullptr, nullptr, nullptr, bits, fa.est);
  }
  // publish: every store of this wave drained first
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store((gu32*)(fa.seq + stream), want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- per-stream control (AspAecBatch_ProcessV / _InitStream): every stream has its own descriptors, read from
// device memory by the wave that owns the stream (wave-uniform: through the scalar cache)
__global__ __launch_bounds__(256) void aec_farend_v_kernel(float* __restrict__ state, float* __restrict__ far_ring,
                                                           const AecTables* __restrict__ G,
                                                           const float* __restrict__ farend, int num_streams,
                                                           const FarOps* __restrict__ vfar, int nrOfSamples, int state_dwords) {
  __shared__ SharedTables T;
  __shared__ float lds[4 * kLdsWave];
  stage_tables(T, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wave;
  if (stream >= num_streams) return;
  typedef const __attribute__((address_space(4))) FarOps* OpsPtr;
  const OpsPtr ops = (OpsPtr)(vfar + stream);
  float* wl = lds + wave * kLdsWave;
  float* st = state + (size_t)stream * state_dwords;
  // the caller's frame holds nrOfSamples per stream; a stream's own `n` is that or 0 (nothing to append)
  farend_work(st, far_ring, wl, T, farend + (size_t)stream * (nrOfSamples - ops->n), num_streams, stream, *ops, lane);
}

template <int NP>
__global__ __launch_bounds__(256, NP == kNumPartNormal ? AEC_WAVES : 3) void aec_process_v_kernel(
    float* __restrict__ state, float* far_ring, const AecTables* __restrict__ G, const float* __restrict__ nearend,
    float* __restrict__ out, int num_streams, int nrOfSamples, const AecStreamStep* __restrict__ vdesc) {
  __shared__ SharedTables T;
  constexpr int kWaveLds = lds_wave(NP, false);
  __shared__ float lds[4 * kWaveLds];
  stage_tables(T, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wave;
  if (stream >= num_streams) return;
  typedef const __attribute__((address_space(4))) AecStreamStep* StepPtr;
  const StepPtr d = (StepPtr)(vdesc + stream);
  float* wl = lds + wave * kWaveLds;
  float* st = state + (size_t)stream * AecRows(NP).state_dwords;
  const float* nin = nearend + (size_t)stream * nrOfSamples;
  float* o = out + (size_t)stream * nrOfSamples;
  if (d->mode == 0) {  // start-up: the near end passes through (echo_cancellation.c:660-664, 768-776)
    for (int i = lane; i < nrOfSamples; i += 64) {
      const float v = nin[i];
      o[i] = v;
    }
    return;
  }
  const FarOps none = {};
  process_call<false, NP, false>(st, far_ring, wl, T, G, nin, o, num_streams, nrOfSamples, stream, lane, d->ops, nullptr, none,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// aec_rdft_forward_128 / inverse_128 seam: four rows per wave.
__global__ __launch_bounds__(256) void aec_rdft128_kernel(const float* __restrict__ src,
                                                          float* __restrict__ dst, int isgn,
                                                          int count,
                                                          const AecTables* __restrict__ G) {
  __shared__ SharedTables T;
  __shared__ float lds[4 * kLdsWave];
  stage_tables(T, G);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: the stream's state block and LDS region get scalar bases
  float* wl = lds + wave * kLdsWave;
  const int row0 = (blockIdx.x * 4 + wave) * 4;
  for (int f = 0; f < 4; ++f) {
    const int row = row0 + f;
    float2 v = {0.f, 0.f};
    if (row < count) {
      v.x = src[(size_t)row * 128 + 2 * lane];
      v.y = src[(size_t)row * 128 + 2 * lane + 1];
    }
    tile(wl, f)[lane] = v;
  }
  wave_fence();
  if (isgn >= 0) {
    rdft_fwd_quad(wl, lane, T);
  } else {
    rdft_inv_quad(wl, lane, T);
  }
  for (int f = 0; f < 4; ++f) {
    const int row = row0 + f;
    if (row < count) {
      const float2 v = tile(wl, f)[lane];
      dst[(size_t)row * 128 + 2 * lane] = v.x;
      dst[(size_t)row * 128 + 2 * lane + 1] = v.y;
    }
  }
}

}  // namespace

namespace aspaec {

hipError_t launch_aec_farend(float* state, float* far_ring, const AecTables* T, const float* farend,
                             int num_streams, const FarOps& ops, hipStream_t s, int stream0, int stream_end,
                             int num_part) {
  if (stream_end < 0) stream_end = num_streams;
  hipLaunchKernelGGL(aec_farend_kernel, dim3((stream_end - stream0 + 3) / 4), dim3(256), 0, s, stream0,
                     stream_end, state, far_ring, T, farend, num_streams, ops, AecRows(num_part).state_dwords);
  return hipGetLastError();
}

hipError_t launch_aec_process(float* state, float* far_ring, const AecTables* T,
                              const float* nearend, float* out, int num_streams, int nrOfSamples,
                              const ProcOps& ops, const float* farend, const FarOps& fops,
                              const float* near_high, float* out_high, float* metrics,
                              hipStream_t s, unsigned long long* stamps, int stream0, int stream_end,
                              int num_part, float* spectra, const DelayBlock* dblocks) {
  if (stream_end < 0) stream_end = num_streams;
  if (num_part != kNumPartNormal && num_part != kNumPartMax) return hipErrorInvalidValue;
  const dim3 grid((stream_end - stream0 + 3) / 4);
#define ASP_AEC_LAUNCH(MET, NP)                                                                              \
  hipLaunchKernelGGL((aec_process_kernel<MET, NP>), grid, dim3(256), 0, s, stream0, stream_end, state,        \
                     far_ring, T, nearend, out, num_streams, nrOfSamples, ops, farend, fops, near_high,      \
                     out_high, metrics, stamps, spectra, dblocks)
  if (num_part == kNumPartNormal) {
    if (metrics != nullptr) ASP_AEC_LAUNCH(true, kNumPartNormal); else ASP_AEC_LAUNCH(false, kNumPartNormal);
  } else {
    if (metrics != nullptr) ASP_AEC_LAUNCH(true, kNumPartMax); else ASP_AEC_LAUNCH(false, kNumPartMax);
  }
#undef ASP_AEC_LAUNCH
  return hipGetLastError();
}

hipError_t launch_aec_process_agn(float* state, float* far_ring, const AecTables* T, const float* nearend, float* out,
                                  int num_streams, int nrOfSamples, const ProcOps& ops, const float* farend, const FarOps& fops,
                                  DelayBlock* dblocks, const AgnOps& agn, hipStream_t s, int stream0, int stream_end,
                                  int num_part) {
  if (num_part != kNumPartNormal && num_part != kNumPartMax) return hipErrorInvalidValue;
  if (stream_end < 0) stream_end = num_streams;
  const dim3 grid((stream_end - stream0 + 3) / 4);
  if (num_part == kNumPartNormal)
    hipLaunchKernelGGL((aec_process_agn_kernel<kNumPartNormal>), grid, dim3(256), 0, s, stream0, stream_end, state, far_ring, T,
                       nearend, out, num_streams, nrOfSamples, ops, farend, fops, dblocks, agn);
  else
    hipLaunchKernelGGL((aec_process_agn_kernel<kNumPartMax>), grid, dim3(256), 0, s, stream0, stream_end, state, far_ring, T,
                       nearend, out, num_streams, nrOfSamples, ops, farend, fops, dblocks, agn);
  return hipGetLastError();
}

// `steps` consecutive frame steps of the hand-off build in one launch (grid y = step): steps want .. want + steps - 1
// of every stream; `descs` [steps] in device memory.
hipError_t launch_aec_process_flow(float* state, float* far_ring, const AecTables* T, int num_streams, int nrOfSamples,
                                   const AecFlowStep* descs, int steps, unsigned* seq, unsigned* abort_w, unsigned want,
                                   int num_part, hipStream_t s, DelayBlock* est, unsigned* bits, bool agn) {
  if (num_part != kNumPartNormal && num_part != kNumPartMax) return hipErrorInvalidValue;
  const int gx = ((num_streams + 3) / 4 + 7) / 8 * 8;  // a multiple of 8: a stream's consecutive steps on one XCD's in-order share
  const dim3 grid(gx, steps);
  const AecFlowArgs fa = {descs, seq, abort_w, want, est, bits};
  // agn: `descs` is an array of AecFlowStepAgn (the delay-agnostic mode: control steps and estimator in the wave)
  if (agn) {
    if (num_part == kNumPartNormal)
      hipLaunchKernelGGL((aec_process_flow_kernel<kNumPartNormal, true>), grid, dim3(256), 0, s, state, far_ring, T, num_streams,
                         nrOfSamples, fa);
    else
      hipLaunchKernelGGL((aec_process_flow_kernel<kNumPartMax, true>), grid, dim3(256), 0, s, state, far_ring, T, num_streams,
                         nrOfSamples, fa);
  } else if (num_part == kNumPartNormal)
    hipLaunchKernelGGL((aec_process_flow_kernel<kNumPartNormal>), grid, dim3(256), 0, s, state, far_ring, T, num_streams,
                       nrOfSamples, fa);
  else
    hipLaunchKernelGGL((aec_process_flow_kernel<kNumPartMax>), grid, dim3(256), 0, s, state, far_ring, T, num_streams,
                       nrOfSamples, fa);
  return hipGetLastError();
}

// per-stream control: one far-end / Process descriptor per stream, in device memory
hipError_t launch_aec_farend_v(float* state, float* far_ring, const AecTables* T, const float* farend, int num_streams,
                               const FarOps* vfar, int nrOfSamples, int num_part, hipStream_t s) {
  hipLaunchKernelGGL(aec_farend_v_kernel, dim3((num_streams + 3) / 4), dim3(256), 0, s, state, far_ring, T, farend, num_streams,
                     vfar, nrOfSamples, AecRows(num_part).state_dwords);
  return hipGetLastError();
}
hipError_t launch_aec_process_v(float* state, float* far_ring, const AecTables* T, const float* nearend, float* out,
                                int num_streams, int nrOfSamples, const AecStreamStep* vdesc, int num_part, hipStream_t s) {
  if (num_part != kNumPartNormal && num_part != kNumPartMax) return hipErrorInvalidValue;
  const dim3 grid((num_streams + 3) / 4);
  if (num_part == kNumPartNormal)
    hipLaunchKernelGGL((aec_process_v_kernel<kNumPartNormal>), grid, dim3(256), 0, s, state, far_ring, T, nearend, out, num_streams,
                       nrOfSamples, vdesc);
  else
    hipLaunchKernelGGL((aec_process_v_kernel<kNumPartMax>), grid, dim3(256), 0, s, state, far_ring, T, nearend, out, num_streams,
                       nrOfSamples, vdesc);
  return hipGetLastError();
}

hipError_t launch_aec_rdft128(const float* src, float* dst, int isgn, int count, const AecTables* T,
                              hipStream_t s) {
  hipLaunchKernelGGL(aec_rdft128_kernel, dim3((count + 15) / 16), dim3(256), 0, s, src, dst, isgn,
                     count, T);
  return hipGetLastError();
}

}  // namespace aspaec
