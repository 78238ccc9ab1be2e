// bt_kernels8.hip -- the BlockThresholding macroblock kernel for N = 1024 (one stream-channel per workgroup) and
// N = 256 (four stream-channels per workgroup, template parameter Q4: see the kernel), one wave per STFT frame.
// Replaces, for many independent stream-channels per launch, one whole macroblock (8 hops) of
//   blockThreshold_STFT / _core / _adaptive_block / blockTreshold_compute_thre /
//   blockThreshold_wiener / blockThreshold_inverse_STFT
//   (Denoise/BlockThresholding/src/audioDenoiseBlockTreshold.c:273-539) and
//   kiss_fftr / kiss_fftri (common/kiss_fft/kiss_fftr.c:67-159, kiss_fft.c:21-302).
//
// Mapping (one workgroup of 8 waves per stream-channel macroblock, four workgroup barriers in all):
//   phase A  wave w = frame w: 8 complex points per lane.  Lane L loads the windowed pairs
//            n = L + 64 j (coalesced 8-byte loads), which are exactly the inputs of kiss_fft's first
//            two decimation stages (radix 2, radix 4 with m = 2) -- they run in registers.  The three
//            remaining radix-4 stages (m = 8, 32, 128) each follow one exchange through the wave's own
//            row of the coefficient tile (XOR-swizzled so that the 8-byte writes and reads are free of
//            bank conflicts; no workgroup barrier, a wave's DS traffic is ordered).  The real-FFT split
//            runs in place on that row and leaves the squared normalised real parts of the SURE search
//            in a second table.
//   phase B1 SURE of the 15 dyadic segmentations (.c:354-401): lane = macro-column, block shapes are
//            compile-time constants, the two half-waves take the two halves of a segmentation's blocks
//            (the second continues the first one's running sum, so the adds keep the reference's order),
//            segmentations are dealt to the 8 waves by cost.
//   phase B2 16 lanes per macro-column: argmin, block powers of the chosen segmentation (each sum in the
//            reference's order), Stein attenuation and the empirical Wiener gain applied in place; the 16
//            spare lanes take the DC column and the bins past the last whole macro-column.
//   phase C  wave w = frame w: kiss_fftri's pre-pass straight from the tile into the registers of the
//            first two stages, the same three exchanges, overlap-add with the neighbour frame's half
//            through LDS, coalesced 8-byte stores.
// Every butterfly, split / merge step and block sum performs the float operations of oracle/bt_oracle.c
// in its order: results are bit-identical to it (tests/test_bt_gpu.py).  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "bt_layout.h"
#include "pk_f32.h"

using namespace aspbt;
using namespace asppk;

namespace {

typedef f32x2 cpx;  // x = re, y = im: one 64-bit register pair, so complex arithmetic issues as packed f32

constexpr int N = 1024, NC = 512, HALF = 512, NB = 513, NCOL = 31;
constexpr int ROW = 544;  // complex slots per coefficient row (NB used; 4352 bytes: rows stay 256-byte aligned for the exchanges)
constexpr int SQS = 31;   // floats per row of the squared-real table [r * 16 + cc][column] (odd: the split's writes spread over the banks)

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The six lane terms (byte offsets inside the wave's row, from the host-built table) with the row's
// byte offset inside the coefficient tile added: rows are 256-byte aligned, so XOR-ing a register term
// below 256 bytes into the sum equals XOR-ing it into the slot index.
struct XTerms {
  int w[3], r[3];
};
__device__ __forceinline__ XTerms exchange_terms(const BtTables* __restrict__ Tb, int lane, int row_bytes) {
  XTerms t;
#pragma unroll
  for (int x = 0; x < 3; ++x) {
    t.w[x] = Tb->xterm[2 * x][lane] + row_bytes;
    t.r[x] = Tb->xterm[2 * x + 1][lane] + row_bytes;
  }
  return t;
}
// One exchange through the wave's private LDS row: registers of layout SRC -> layout DST.  swz is linear
// over GF(2) and a position is the XOR of its lane part and its register part; the register part's bits
// above the swizzled five are disjoint from the lane part's, so they add (an immediate offset).
template <int X>
__device__ __forceinline__ void exchange(cpx (&v)[8], cpx* tile, const XTerms& xt) {
  constexpr Lay SRC = X == 1 ? LA : X == 2 ? LB : LC;
  constexpr Lay DST = X == 1 ? LB : X == 2 ? LC : LD;
  constexpr Swz S = X == 1 ? S1 : X == 2 ? S2 : S3;
  // laundered so that the 16 slot addresses are recomputed here (one XOR each) instead of being kept
  // alive -- and spilled -- between the forward and the inverse transform
  int wl = xt.w[X - 1], rl = xt.r[X - 1];
  asm volatile("" : "+v"(wl), "+v"(rl));
  char* base = reinterpret_cast<char*>(tile);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = swz(S, pos_reg(SRC, j));
    *reinterpret_cast<cpx*>(base + ((wl ^ ((c & 31) * 8)) + (c & ~31) * 8)) = v[j];
  }
  wave_lds_fence();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = swz(S, pos_reg(DST, j));
    v[j] = *reinterpret_cast<const cpx*>(base + ((rl ^ ((c & 31) * 8)) + (c & ~31) * 8));
  }
  wave_lds_fence();
}

// kf_bfly2 (kiss_fft.c:21-42), m = 1
template <bool INV>
__device__ __forceinline__ void bfly2(cpx& f0, cpx& f1, cpx tw0) {
  const cpx t = cmul<INV>(f1, tw0);
  f1 = f0 - t;
  f0 = f0 + t;
}
// kf_bfly4 (kiss_fft.c:44-90): a0..a3 = Fout[0], Fout[m], Fout[2m], Fout[3m]; the twiddles are the
// forward table's (INV multiplies by their conjugates = the inverse table's entries)
template <bool INV>
__device__ __forceinline__ void bfly4(cpx& a0, cpx& a1, cpx& a2, cpx& a3, cpx t1, cpx t2, cpx t3) {
  const cpx s0 = cmul<INV>(a1, t1);
  const cpx s1 = cmul<INV>(a2, t2);
  const cpx s2 = cmul<INV>(a3, t3);
  const cpx s5 = a0 - s1;
  cpx f0 = a0 + s1;
  const cpx s3 = s0 + s2;
  const cpx s4 = s0 - s2;
  a2 = f0 - s3;
  a0 = f0 + s3;
  if (INV) {
    a1 = add_swap_sub_lo(s5, s4);  // {s5.r - s4.i, s5.i + s4.r}
    a3 = add_swap_sub_hi(s5, s4);  // {s5.r + s4.i, s5.i - s4.r}
  } else {
    a1 = add_swap_sub_hi(s5, s4);
    a3 = add_swap_sub_lo(s5, s4);
  }
}

// Twiddles: the forward tables (the inverse ones are their conjugates bit for bit -- bt_api.hip checks
// that when it builds them) are staged in LDS once per workgroup (kTwLds entries of kiss_fft's table,
// all the stages below index, and the 256 super twiddles of kiss_fftr) and read just ahead of each stage.
constexpr int kTwLds = 382;  // 3 * 127 is the largest index any stage uses
constexpr int SUS = 15;      // SURE values per macro-column

// All stages of one 512-point complex FFT of kiss_fft (factors 4,4,4,4,2): in: layout LA, out: LD.
// tw: the global table (wave-uniform entries of the first two stages), twl: its LDS copy
// Q4: four independent 128-point transforms (kiss factors 4,4,4,2) of four streams whose inputs are interleaved
// (input 4 n' + s = sample n' of stream s): they are this network without its last stage, and their twiddles are
// entries of the same table (tw128[j] == tw512[4 j] bit for bit: the phases differ by exact powers of two).  The
// last exchange still runs: stream s ends in natural order at positions 128 s .. 128 s + 127 (layout LD).
template <bool INV, bool Q4 = false>
__device__ __forceinline__ void wave_fft512(cpx (&v)[8], cpx* tile, const cpx* twl,
                                            const cpx* __restrict__ tw, int lane, const XTerms& xt) {
  {  // radix-2 leaves (m = 1, twiddle 0) and the radix-4 stage with m = 2: wave-uniform twiddles
    const cpx t0 = tw[0], t64 = tw[64], t128 = tw[128], t192 = tw[192];
    bfly2<INV>(v[0], v[1], t0);
    bfly2<INV>(v[2], v[3], t0);
    bfly2<INV>(v[4], v[5], t0);
    bfly2<INV>(v[6], v[7], t0);
    bfly4<INV>(v[0], v[2], v[4], v[6], t0, t0, t0);
    bfly4<INV>(v[1], v[3], v[5], v[7], t64, t128, t192);
  }
  {
    const int kb = lane & 7;
    const cpx t1 = twl[kb * 16], t2 = twl[kb * 32], t3 = twl[kb * 48];
    exchange<1>(v, tile, xt);
#pragma unroll
    for (int c = 0; c < 2; ++c) bfly4<INV>(v[c], v[c + 2], v[c + 4], v[c + 6], t1, t2, t3);
  }
  {
    const int kc = lane & 31;
    const cpx t1 = twl[kc * 4], t2 = twl[kc * 8], t3 = twl[kc * 12];
    exchange<2>(v, tile, xt);
#pragma unroll
    for (int c = 0; c < 2; ++c) bfly4<INV>(v[c], v[c + 2], v[c + 4], v[c + 6], t1, t2, t3);
  }
  if constexpr (Q4) {
    exchange<3>(v, tile, xt);
  } else {
    cpx t[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 3; ++q) t[c][q] = twl[(lane + 64 * c) * (q + 1)];
    exchange<3>(v, tile, xt);
#pragma unroll
    for (int c = 0; c < 2; ++c) bfly4<INV>(v[c], v[c + 2], v[c + 4], v[c + 6], t[c][0], t[c][1], t[c][2]);
  }
}

// register j of layout LA holds input n = lane + 64 k3 + 256 k4 with j = 2 k3 + k4
__device__ __forceinline__ constexpr int la_input(int j) { return 64 * (j >> 1) + 256 * (j & 1); }

// kiss_fftr post-pass (kiss_fftr.c:92-120), in place on the wave's row (natural order T[0..511]):
// lane takes k = 1 + lane + 64 i; lane 0 also k = 0.  SQ: also leave (re * norm)^2 of every bin of a
// whole macro-column in the squared-real table.
template <bool SQ>
__device__ __forceinline__ void wave_split_forward(cpx* row, const cpx (&sp)[4], int lane,
                                                   float* sq_fr, float norm) {
  cpx fp[4], fn[4];  // sp[i] = super twiddle k - 1 = lane + 64 i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 1 + lane + 64 * i;
    fp[i] = row[k];
    fn[i] = row[NC - k];
  }
  const cpx t0 = row[0];
  wave_lds_fence();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 1 + lane + 64 * i;
    // fpnk = conj(T[NC - k]); f1k = fpk + fpnk; f2k = fpk - fpnk
    const cpx f1k = add_sub_hi(fp[i], fn[i]);
    const cpx f2k = add_sub_lo(fp[i], fn[i]);
    const cpx twv = cmul<false>(f2k, sp[i]);
    const cpx a = (f1k + twv) * 0.5f;
    const cpx b = sub_lo_rsub_hi(f1k, twv) * 0.5f;  // {(f1k.r - tw.r) / 2, (tw.i - f1k.i) / 2}
    if (k != NC - k) row[k] = a;
    row[NC - k] = b;  // for k = NC/2 the second assignment is the one that stays
    if (SQ) {
      // bins 1 .. 16 NCOL belong to macro-columns: bin = 1 + 16 m + cc
      const int ka = k - 1, kb = NC - k - 1;
      f32x2 re = {a.x, b.x};
      re = re * norm;
      re = re * re;
      if (k != NC - k) sq_fr[(ka & 15) * SQS + (ka >> 4)] = re.x;
      if (kb < 16 * NCOL) sq_fr[(kb & 15) * SQS + (kb >> 4)] = re.y;
    }
  }
  if (lane == 0) {
    cpx z;
    z.x = t0.x + t0.y;
    z.y = 0.f;
    row[0] = z;
    z.x = t0.x - t0.y;
    row[NC] = z;
  }
}

// kiss_fftri pre-pass (kiss_fftr.c:137-157) from the row (natural order F[0..512]) straight into the
// registers of layout LA: register j wants T[n], n = lane + la_input(j).
// register j's pair index k: n itself below 256, NC - n above (it is then the (NC - k) side of its pair)
__device__ __forceinline__ int merge_sup_index(int lane, int j) {
  const int n = lane + la_input(j);
  const int k = (j & 1) ? NC - n : n;
  return k > 0 ? k - 1 : 0;
}
__device__ __forceinline__ void wave_merge_inverse(cpx (&v)[8], const cpx* row, const cpx (&spf)[8],
                                                   int lane) {
  cpx u[8], w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = lane + la_input(j);
    u[j] = row[n];
    w[j] = row[NC - n];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const cpx fk = (j & 1) ? w[j] : u[j], fo = (j & 1) ? u[j] : w[j];  // F[k], F[NC - k]
    // fnkc = conj(F[NC - k]); fek = fk + fnkc; tmp = fk - fnkc; fok = tmp * (inverse super twiddle =
    // conjugate of the forward one)
    const cpx fek = add_sub_hi(fk, fo);
    const cpx tmp = add_sub_lo(fk, fo);
    const cpx fok = cmul<true>(tmp, spf[j]);
    // n < 256: T[k] = fek + fok; above: T[NC - k] = {fek.r - fok.r, (fek.i - fok.i) * -1}
    cpx t = (j & 1) ? sub_lo_rsub_hi(fek, fok) : fek + fok;
    if (j == 0 && lane == 0) {  // n = 0
      t.x = u[0].x + w[0].x;
      t.y = u[0].x - w[0].x;
    }
    v[j] = t;
  }
}

// ---- four streams of 256-sample windows per workgroup (Q4): the row holds T_s[k] at 128 s + k after the
// transform; the spectra F_s[0 .. 128] go to 129 s + b.  Pairs (k, 128 - k): lane takes k = 1 + lane of every
// stream; lanes 0..3 also bin 0 / 128 of stream `lane`.
constexpr int NCQ = 128, NBQ = 129, NCOLQ = 7;  // per stream: complex points, bins, whole macro-columns
template <bool SQ>
__device__ __forceinline__ void wave_split_forward_q4(cpx* row, cpx spq, int lane, float* sq_fr, float norm) {
  cpx fp[4], fn[4];
  const int k = 1 + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    fp[q] = row[NCQ * q + k];
    fn[q] = row[NCQ * q + NCQ - k];
  }
  const cpx t0 = row[NCQ * (lane & 3)];
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const cpx f1k = add_sub_hi(fp[q], fn[q]);
    const cpx f2k = add_sub_lo(fp[q], fn[q]);
    const cpx twv = cmul<false>(f2k, spq);
    const cpx a = (f1k + twv) * 0.5f;
    const cpx b = sub_lo_rsub_hi(f1k, twv) * 0.5f;
    if (k != NCQ - k) row[NBQ * q + k] = a;
    row[NBQ * q + NCQ - k] = b;  // for k = 64 the second assignment is the one that stays
    if (SQ) {
      const int ka = k - 1, kb = NCQ - k - 1;  // bin = 1 + 16 m' + cc of column 7 q + m'
      f32x2 re = {a.x, b.x};
      re = re * norm;
      re = re * re;
      if (k != NCQ - k) sq_fr[(ka & 15) * SQS + NCOLQ * q + (ka >> 4)] = re.x;
      if (kb < 16 * NCOLQ) sq_fr[(kb & 15) * SQS + NCOLQ * q + (kb >> 4)] = re.y;
    }
  }
  if (lane < 4) {
    cpx z;
    z.x = t0.x + t0.y;
    z.y = 0.f;
    row[NBQ * lane] = z;
    z.x = t0.x - t0.y;
    row[NBQ * lane + NCQ] = z;
  }
}

// register j of layout LA holds input 4 n' + s of the interleaved array: s = lane & 3,
// n' = (lane >> 2) + 16 (j >> 1) + 64 (j & 1)
__device__ __forceinline__ constexpr int q4_input(int lane, int j) { return (lane >> 2) + 16 * (j >> 1) + 64 * (j & 1); }
__device__ __forceinline__ int merge_sup_index_q4(int lane, int j) {
  const int n = q4_input(lane, j);
  const int k = (j & 1) ? NCQ - n : n;
  return k > 0 ? k - 1 : 0;
}
__device__ __forceinline__ void wave_merge_inverse_q4(cpx (&v)[8], const cpx* row, const cpx (&spf)[8], int lane) {
  const cpx* F = row + NBQ * (lane & 3);
  cpx u[8], w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = q4_input(lane, j);
    u[j] = F[n];
    w[j] = F[NCQ - n];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const cpx fk = (j & 1) ? w[j] : u[j], fo = (j & 1) ? u[j] : w[j];  // F[k], F[128 - k]
    const cpx fek = add_sub_hi(fk, fo);
    const cpx tmp = add_sub_lo(fk, fo);
    const cpx fok = cmul<true>(tmp, spf[j]);
    cpx t = (j & 1) ? sub_lo_rsub_hi(fek, fok) : fek + fok;
    if (j == 0 && lane < 4) {  // n' = 0
      t.x = u[0].x + w[0].x;
      t.y = u[0].x - w[0].x;
    }
    v[j] = t;
  }
}

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
template <int TT, int FF>
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
template <int T, int F>
__device__ __forceinline__ void sure_seg(const float* sq, float* sure, const BtSeg* __restrict__ sgp, int lane) {
  const float temp = sgp->temp, size_blk = sgp->size_blk, thr = sgp->thr, two_size = sgp->two_size;
  constexpr int TT = 8 >> T, FF = 16 >> F, S = T + F, c = T * 5 + F;
  const int h = lane >> 5, m = lane & 31;
  if constexpr (S == 0) {
    float t[1];
    t[0] = block_sum<8, 16>(sq + m, 0, 0);
    sure_terms<1>(t, temp, size_blk, thr, two_size);
    float s = 0.0f;
    s += t[0];
    if (lane < NCOL) sure[m * SUS + c] = s;
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
      for (int q = 0; q < CH; ++q) x[q] = block_sum<TT, FF>(col, TT * ((c0 + q) / NJ), FF * ((c0 + q) % NJ));
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
    if (h == 1 && m < NCOL) sure[m * SUS + c] = s1;
  }
}

// The segmented running sums of phase B2 for one DPP row (= one macro-column): see the kernel.  STEPS =
// (widest segment of the wave's columns) - 1, unrolled.  One `v_add_f32_dpp acc, acc, pw row_shr:1` per
// step, with no select: a lane whose DPP source is invalid -- outside the row, or switched off in EXEC --
// keeps its value (bound_ctrl 0).  With the LAST lane of every segment switched off, the first lane of the
// next segment keeps its start value (carry + own power) while the sums run along the segment; one more
// step with every lane on gives the last lanes their sums (it spoils the first lanes, which are done: the
// next row starts them afresh; one-lane segments take their start value).  Lanes that already hold their
// final sum recompute the same value.  The DPP reads follow VALU writes of the same register: two wait
// states each, given here because the hazard recogniser does not look into inline assembly.
#define BT8_DPP1 "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define BT8_DPP2 BT8_DPP1 BT8_DPP1
#define BT8_DPP6 BT8_DPP2 BT8_DPP2 BT8_DPP2
#define BT8_DPP14 BT8_DPP6 BT8_DPP6 BT8_DPP2
#define BT8_DPP_STEPS(reps, acc, p) asm volatile(reps : "+v"(acc) : "v"(p))
template <int STEPS>
__device__ __forceinline__ void scan_rows(const float (&pw)[8], bool seglast, bool single, int lastlane, int TT,
                                          float& tot1, float& tot3, float& tot5, float& tot7) {
  float carry = 0.0f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float base = ((r & (TT - 1)) == 0 ? 0.0f : carry) + pw[r];
    float acc = base;
    if constexpr (STEPS >= 2) {
      if (!seglast) {  // one block per row: nothing of the compiler's between the steps
        static_assert(STEPS == 15 || STEPS == 7 || STEPS == 3 || STEPS < 2, "segment widths are powers of two");
        if constexpr (STEPS == 15) BT8_DPP_STEPS(BT8_DPP14, acc, pw[r]);
        if constexpr (STEPS == 7) BT8_DPP_STEPS(BT8_DPP6, acc, pw[r]);
        if constexpr (STEPS == 3) BT8_DPP_STEPS(BT8_DPP2, acc, pw[r]);
      }
    }
    if constexpr (STEPS >= 1) {
      BT8_DPP_STEPS(BT8_DPP1, acc, pw[r]);
      acc = single ? base : acc;
    }
    carry = __int_as_float(__builtin_amdgcn_ds_bpermute(lastlane, __float_as_int(acc)));
    if (r == 1) tot1 = carry;
    if (r == 3) tot3 = carry;
    if (r == 5) tot5 = carry;
    if (r == 7) tot7 = carry;
  }
}

// The same running sums when every macro-column of the wave has chosen segments 16 bins wide (F = 0; the
// usual choice in stationary noise): a block sum then runs along the whole DPP row and on into the next
// row, so a row is one rotate-add (lane 0 takes lane 15's sum of the row before) and 15 shift-adds whose
// lane 0 keeps its value (row_shr with no source lane leaves the destination alone): 128 dependent adds
// per column, the reference's own count, with no select and no LDS hop on the chain.  The DPP reads follow
// VALU writes of the same register: two wait states each, given here because the hazard recogniser does
// not look into inline assembly.  Lanes of the DC / tail row (`edge`: one-lane blocks of 8 rows) add
// their own eight powers in order.
__device__ __forceinline__ void scan_rows_full(const float (&pw)[8], int lastlane, int TT, bool edge,
                                               float& tot1, float& tot3, float& tot5, float& tot7) {
  float acc = pw[0], own = pw[0];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (r > 0) {
      own += pw[r];
      float nxt;
      asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf" : "=&v"(nxt) : "v"(acc), "v"(pw[r]));
      const bool starts = (r & 1) == 0 && (r & (TT - 1)) == 0;  // a block starts at rows 2 / 4 / 6 for TT = 2 / <= 4 / 2
      acc = starts ? pw[r] : nxt;
    }
    BT8_DPP_STEPS(BT8_DPP14 BT8_DPP1, acc, pw[r]);
    if (r & 1) {
      const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(lastlane, __float_as_int(acc)));
      if (r == 1) tot1 = t;
      if (r == 3) tot3 = t;
      if (r == 5) tot5 = t;
      if (r == 7) tot7 = edge ? own : t;
    }
  }
}

// per-wave stamps (diagnostic): slots 16 + 4 wave + {0: end of phase A, 1: end of SURE, 2: end of phase B2, 3: end}
#define BT8_WSTAMP(k) \
  if (stamps != nullptr && blockIdx.x == 0 && lane == 0) stamps[16 + 4 * wave + (k)] = __builtin_amdgcn_s_memtime();
#define BT8_STAMP(k) \
  if (stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime();

#ifndef BT8_WAVES
#define BT8_WAVES 4  // waves per SIMD the register allocation aims at (measured: 6 = three workgroups per CU is 4 % slower, it spills)
#endif
// Q4: four stream-channels of 256-sample windows per workgroup (workgroup g takes streams 4 g .. 4 g + 3): wave w =
// frame w of all four, their samples interleaved on the way in so that the 512-point network computes four 128-point
// transforms; 4 x 7 macro-columns + 4 x 16 edge lanes are the 512 threads of phase B2.
// The hand-off build (FLOW): one launch carries M consecutive macroblocks of every stream-channel (blockIdx.y = the
// step; its input / output is ring slot (slot0 + step) % ring).  What a stream-channel's consecutive macroblocks hand
// each other is small: the input tail IS the last half window of the previous macroblock's input (read from there,
// no hand-off; only the launch's first step reads the state's copy and only its last step writes it), and the
// overlap-add tail, written by wave 7 at the very end of a macroblock and read by wave 0 of the next one just before
// its own overlap-add -- ordered by a per-stream step counter in memory (sc1 stores drained, then the counter; wave 0
// polls it, then sc1 loads), as in ns_kernels1.hip.  Workgroups are dispatched in grid order, so the macroblock
// waited for has been dispatched a whole step earlier.  A wait that times out sets the abort word and the workgroup
// carries on with a zero tail (the barriers that follow need every wave); bt_api.hip reports it.
struct BtFlowArgs {
  unsigned* seq;      // [num_streams]: macroblocks stream s has completed in hand-off launches
  unsigned* abort_w;
  unsigned want;      // blockIdx.y == 0 is macroblock `want` of every stream
  int slot0, ring;
  unsigned per;       // floats between two ring slots of in / out
};
typedef __attribute__((address_space(1))) unsigned bt_gu32;
typedef __attribute__((address_space(1))) unsigned long long bt_gu64;
__device__ __forceinline__ f32x2 ld2_sc1(const float* p) {
  const unsigned long long v = __hip_atomic_load((const bt_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(f32x2, v);
}
__device__ __forceinline__ void st2_sc1(float* p, f32x2 v) {
  __hip_atomic_store((bt_gu64*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool Q4, bool FLOW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BT8_WAVES, BT8_WAVES))) void bt_macroblock8_kernel(
    float* __restrict__ state, const BtTables* __restrict__ Tb, const float* __restrict__ in,
    float* __restrict__ out, int in_stride, int out_stride, unsigned long long* __restrict__ stamps, BtFlowArgs fa) {
  __shared__ __align__(256) cpx coef[8 * ROW];     // [frame][bin]; a wave's row is also its exchange buffer
  __shared__ __align__(16) float sq[128 * SQS];    // squared normalised real parts; later the OLA halves
  __shared__ float sure[NCOL * SUS + 15];          // [macro-column][segmentation], then a_const of the 15 segmentations
  __shared__ __align__(16) cpx twl[kTwLds];        // kiss_fft twiddles 0..383 (forward)
  BT8_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the phase switches are scalar branches
  constexpr int HALFV = Q4 ? 128 : HALF, NCOLV = Q4 ? 4 * NCOLQ : NCOL;
  // the stream whose samples this lane loads (Q4: input 4 n' + s of the interleaved array sits on lanes with lane & 3 = s)
  const int stream = Q4 ? 4 * blockIdx.x + (lane & 3) : blockIdx.x;
  float* st = state + (size_t)stream * kStateFloats;
  const float* in_prev = in;  // hand-off build: the previous macroblock's input (steps after the launch's first)
  unsigned flow_want = 0;
  bool flow_first = true, flow_last = true;
  if constexpr (FLOW) {
    const unsigned j = blockIdx.y;
    const unsigned slot = ((unsigned)fa.slot0 + j) % (unsigned)fa.ring;
    const unsigned pslot = ((unsigned)fa.slot0 + j + (unsigned)fa.ring - 1u) % (unsigned)fa.ring;
    in_prev = in + (size_t)pslot * fa.per;
    in += (size_t)slot * fa.per;
    out += (size_t)slot * fa.per;
    flow_want = fa.want + j;
    flow_first = j == 0;
    flow_last = j + 1 == gridDim.y;
  }
  const float* x = in + (size_t)stream * in_stride;
  float* y = out + (size_t)stream * out_stride;
  const BtSize& P = Q4 ? Tb->s256 : Tb->s1024;
  const cpx* tw = reinterpret_cast<const cpx*>(Tb->tw1024_f);
  const cpx* sup = reinterpret_cast<const cpx*>(Q4 ? Tb->sup256_f : Tb->sup1024_f);
  cpx* row = coef + wave * ROW;
  cpx* tile = coef;
  const XTerms xt = exchange_terms(Tb, lane, wave * ROW * 8);

  // ---------------------------------------------------------------- phase A: STFT of frame `wave`
  // (blockThreshold_STFT, .c:273-282): frame t sees B[HALF t .. HALF t + N) of B = [inbuf tail | new samples]
  cpx v[8];
  cpx spf[4];  // the split's super twiddles, requested with the samples
#pragma unroll
  for (int i = 0; i < 4; ++i) spf[i] = sup[Q4 ? lane : lane + 64 * i];
  {
    const f32x2* hann2 = reinterpret_cast<const f32x2*>(Q4 ? Tb->hann256 : Tb->hann1024);
    f32x2 s[8], hw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = Q4 ? q4_input(lane, j) : lane + la_input(j);  // complex input index: samples 2n, 2n + 1 of the frame
      // frame 0's first half window: the carried input tail -- the state's copy, or (hand-off build, not the
      // launch's first step) the last half window of the previous macroblock's input, which is what that copy holds
      const float* tail_src = (FLOW && !flow_first) ? in_prev + (size_t)stream * in_stride + 7 * HALFV + 2 * n
                                                    : st + kOffInTail + 2 * n;
      const float* src = (wave == 0 && (j & 1) == 0) ? tail_src : x + HALFV * (wave - 1) + 2 * n;
      s[j] = *reinterpret_cast<const f32x2*>(src);
      hw[j] = hann2[n];
    }
    if (tid < kTwLds) twl[tid] = tw[tid];
    float* segc = sure + NCOL * SUS;
    if (tid < 15) segc[tid] = P.seg[tid / 5][tid % 5].a_const;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = s[j] * hw[j];
    }
    __syncthreads();  // the twiddle tables are staged (every wave is waiting for its samples here anyway)
    BT8_STAMP(1)
    wave_fft512<false, Q4>(v, tile, twl, tw, lane, xt);
  }
  BT8_STAMP(2)
#pragma unroll
  for (int j = 0; j < 8; ++j) row[lane + 64 * j] = v[j];
  wave_lds_fence();
  if constexpr (Q4)
    wave_split_forward_q4<true>(row, spf[0], lane, sq + wave * 16 * SQS, P.norm);
  else
    wave_split_forward<true>(row, spf, lane, sq + wave * 16 * SQS, P.norm);
  BT8_STAMP(3)
  BT8_WSTAMP(0)
  __syncthreads();
  BT8_STAMP(4)
  // carry the last HALF input samples (wave 0 has read the old tail before the barrier); hand-off build: only the
  // launch's last step (the steps in between read them from their predecessor's input)
  if (!flow_last) {
  } else if constexpr (Q4) {
    const size_t sq4 = (size_t)4 * blockIdx.x + (tid >> 7);
    state[sq4 * kStateFloats + kOffInTail + (tid & 127)] = in[sq4 * in_stride + 7 * HALFV + (tid & 127)];
  } else {
    st[kOffInTail + tid] = x[7 * HALF + tid];
  }

  // ---------------------------------------------------------------- phase B1: SURE (.c:354-401)
  switch (wave) {  // segmentations dealt by cost (terms per half-wave: 32, 16, 16, 8, ...)
    case 0: sure_seg<2, 4>(sq, sure, &P.seg[2][4], lane); break;
    case 1: sure_seg<2, 3>(sq, sure, &P.seg[2][3], lane); sure_seg<0, 1>(sq, sure, &P.seg[0][1], lane); break;
    case 2: sure_seg<1, 4>(sq, sure, &P.seg[1][4], lane); sure_seg<1, 0>(sq, sure, &P.seg[1][0], lane); break;
    case 3: sure_seg<2, 2>(sq, sure, &P.seg[2][2], lane); sure_seg<0, 0>(sq, sure, &P.seg[0][0], lane); break;
    case 4: sure_seg<1, 3>(sq, sure, &P.seg[1][3], lane); sure_seg<2, 1>(sq, sure, &P.seg[2][1], lane); break;
    case 5: sure_seg<0, 4>(sq, sure, &P.seg[0][4], lane); sure_seg<1, 2>(sq, sure, &P.seg[1][2], lane); break;
    case 6: sure_seg<0, 3>(sq, sure, &P.seg[0][3], lane); sure_seg<0, 2>(sq, sure, &P.seg[0][2], lane); break;
    default: sure_seg<2, 0>(sq, sure, &P.seg[2][0], lane); sure_seg<1, 1>(sq, sure, &P.seg[1][1], lane); break;
  }
  BT8_STAMP(5)
  BT8_WSTAMP(1)
  __syncthreads();
  BT8_STAMP(6)

  // ---------------------------------------------------------------- phase B2: attenuation + Wiener, in place
  {
    const int m = tid >> 4, u = tid & 15;
    {
      // Lanes 496..511 (m == NCOL) own the DC column and the bins past the last whole macro-column
      // (.c:501-506, 518-532): one column per lane, a block of the 8 frames, and a gain formed in float.
      // They run the columns' code as a (T, F) = (0, 4) block (one lane wide, 8 rows: the same sums in
      // the same order) and swap the gain in at the end.  The Nyquist bin (512) is thresholded by the
      // reference but never reaches the Wiener step: it stays as it is.
      const bool edge = m >= NCOLV;
      // argmin, first minimum wins (.c:404-416)
      float best = sure[m * SUS];
      int bc = 0;
#pragma unroll
      for (int c = 1; c < 15; ++c) {
        const float s = sure[m * SUS + c];
        if (s < best) {
          best = s;
          bc = c;
        }
      }
      BT8_STAMP(11)
      if (edge) bc = 4;
      const int T = bc >= 10 ? 2 : bc >= 5 ? 1 : 0, F = bc - 5 * T;
      const int TT = 8 >> T, FF = 16 >> F;
      const float a_const = sure[NCOL * SUS + bc];
      cpx* col;
      if constexpr (Q4) {  // column m = 7 q + m' of stream q; edge group 28 + q: bin 0 and bins 113..127 of stream q
        const int q = edge ? m - NCOLV : m / NCOLQ;
        col = coef + NBQ * q + (edge ? (u == 0 ? 0 : 16 * NCOLQ + u) : 1 + 16 * (m - NCOLQ * q) + u);
      } else {
        col = coef + (edge ? (u == 0 ? 0 : 16 * NCOL + u) : 1 + 16 * m + u);
      }
      // Block powers of the chosen segmentation (.c:421-454).  The column's 16 lanes are one DPP row: lane u
      // owns bin u of every frame.  A block sum runs rows outer / columns inner, i.e. along the lanes of a
      // segment of FF lanes and on into the next row: a segmented left-to-right scan per row (each step adds
      // the left neighbour's running sum, so the adds are the reference's, in its order), the row's last
      // lane handing the sum to the next row's first lane.
      cpx z[8];
      float pw[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) z[r] = col[r * ROW];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const f32x2 zz = z[r] * z[r];
        pw[r] = zz.x + zz.y;
      }
      BT8_STAMP(12)
      const bool seglast = (u & (FF - 1)) == FF - 1, single = FF == 1;
      const int lastlane = (lane | (FF - 1)) << 2;
      int mf = FF;
      mf = max(mf, __shfl_xor(mf, 16));
      mf = max(mf, __shfl_xor(mf, 32));
      const int steps = __builtin_amdgcn_readfirstlane(mf) - 1;
      float tot1 = 1.0f, tot3 = 1.0f, tot5 = 1.0f, tot7 = 1.0f;  // running sums after rows 1, 3, 5, 7
      const bool full = steps == 15 && __builtin_expect(__all(edge || FF == 16), 1);
      if (full) {
        scan_rows_full(pw, lastlane, TT, edge, tot1, tot3, tot5, tot7);
      } else switch (steps) {
        case 15: scan_rows<15>(pw, seglast, single, lastlane, TT, tot1, tot3, tot5, tot7); break;
        case 7: scan_rows<7>(pw, seglast, single, lastlane, TT, tot1, tot3, tot5, tot7); break;
        case 3: scan_rows<3>(pw, seglast, single, lastlane, TT, tot1, tot3, tot5, tot7); break;
        case 1: scan_rows<1>(pw, seglast, single, lastlane, TT, tot1, tot3, tot5, tot7); break;
        default: scan_rows<0>(pw, seglast, single, lastlane, TT, tot1, tot3, tot5, tot7); break;
      }
      BT8_STAMP(13)
      // a block ends at row r when (r + 1) % TT == 0; its Stein gain (.c:440-444).  Rows 1 and 5 end blocks
      // only for TT = 2, row 3 for TT <= 4, row 7 always; sums that end no block are replaced by 1
      if (TT != 2) tot1 = 1.0f, tot5 = 1.0f;
      if (TT == 8) tot3 = 1.0f;
      const float pmin = fminf(fminf(tot1, tot3), fminf(tot5, tot7)), pmax = fmaxf(fmaxf(tot1, tot3), fmaxf(tot5, tot7));
      float g1 = (float)(1.0 - (double)fdiv_lean(a_const, tot1));
      float g3 = (float)(1.0 - (double)fdiv_lean(a_const, tot3));
      float g5 = (float)(1.0 - (double)fdiv_lean(a_const, tot5));
      float g7 = (float)(1.0 - (double)fdiv_lean(a_const, tot7));
      if (__builtin_expect(!(pmin >= 1e-18f && pmax <= 1e18f), 0)) {
        g1 = (float)(1.0 - (double)(a_const / tot1));
        g3 = (float)(1.0 - (double)(a_const / tot3));
        g5 = (float)(1.0 - (double)(a_const / tot5));
        g7 = (float)(1.0 - (double)(a_const / tot7));
      }
      g1 = g1 * (float)(g1 > 0);
      g3 = g3 * (float)(g3 > 0);
      g5 = g5 * (float)(g5 > 0);
      g7 = g7 * (float)(g7 > 0);
      if (edge) {  // (.c:503-505, 523-525)
        g7 = 1 - fdiv_checked(P.dc_const, tot7);
        if (g7 < 0) g7 = 0;
      }
      BT8_STAMP(14)
      // thresholded coefficient -> empirical Wiener gain on the original one (.c:446-452, 469-486)
      float wn[8], den[8], wmin, wmax;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        // the block of row r ends at row r | (TT - 1): 7 (T = 0), 3 or 7 (T = 1), r | 1 (T = 2)
        const float gpair = r < 2 ? g1 : r < 4 ? g3 : r < 6 ? g5 : g7;
        const float a = T == 2 ? gpair : (T == 1 && r < 4) ? g3 : g7;
        f32x2 th = z[r] * a;
        th = th * th;
        wn[r] = th.x + th.y;
        den[r] = wn[r] + P.wiener_c;
        wmin = r == 0 ? wn[r] : fminf(wmin, wn[r]);
        wmax = r == 0 ? wn[r] : fmaxf(wmax, wn[r]);
      }
      // lean divisions are exact for 0 and for [1e-30, 1e18]; anything else takes the IEEE form (rare)
      const bool lean = wmax <= 1e18f && (wmin >= 1e-30f || wmin == 0.0f);
      const bool all_lean = __builtin_expect(__all(lean), 1);
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        f32x2 gw;
        if (all_lean) {
          gw = fdiv2_lean(f32x2{wn[r], wn[r + 1]}, f32x2{den[r], den[r + 1]});
        } else {
          gw.x = wn[r] / den[r];
          gw.y = wn[r + 1] / den[r + 1];
        }
        col[r * ROW] = z[r] * gw.x;
        col[(r + 1) * ROW] = z[r + 1] * gw.y;
      }
    }
  }
  // the old output tail, for wave 0's overlap-add (requested ahead of the barrier)
  // (Q4: after the inverse transform register pair (2 q, 2 q + 1) is stream q: samples 2 lane, 2 lane + 1 of the
  // frame's first / second half)
  const size_t g4 = (size_t)4 * blockIdx.x;
  f32x2 tail[4];
  if (wave == 0) {
    if constexpr (FLOW) {
      // the stream-channel's previous macroblock must have published its overlap-add tail (Q4: four stream-channels)
      const bt_gu32* f = (const bt_gu32*)(fa.seq + (Q4 ? g4 + (lane & 3) : (size_t)blockIdx.x));
      bool ok = true;
      for (unsigned spins = 0;;) {
        const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v == flow_want)) break;
        ++spins;
        if ((spins & 63u) == 0u &&
            __builtin_amdgcn_readfirstlane((int)__hip_atomic_load((const bt_gu32*)fa.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) {
          ok = false;
          break;
        }
        if (spins > (1u << 17)) {
          if (lane == 0) __hip_atomic_store((bt_gu32*)fa.abort_w, 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tail[j] = f32x2{0.0f, 0.0f};
        if (ok)
          tail[j] = Q4 ? ld2_sc1(state + (g4 + j) * kStateFloats + kOffOutTail + 2 * lane)
                       : ld2_sc1(st + kOffOutTail + 2 * (lane + 64 * j));
      }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      tail[j] = Q4 ? *reinterpret_cast<const f32x2*>(state + (g4 + j) * kStateFloats + kOffOutTail + 2 * lane)
                   : *reinterpret_cast<const f32x2*>(st + kOffOutTail + 2 * (lane + 64 * j));
    }
  }
  cpx spi[8];  // the merge's super twiddles, requested ahead of the barrier
#pragma unroll
  for (int j = 0; j < 8; ++j) spi[j] = sup[Q4 ? merge_sup_index_q4(lane, j) : merge_sup_index(lane, j)];
  BT8_STAMP(7)
  BT8_WSTAMP(2)
  __syncthreads();
  BT8_STAMP(8)

  // ---------------------------------------------------------------- phase C: inverse STFT + overlap-add
  // (blockThreshold_inverse_STFT, .c:284-300)
  if constexpr (Q4)
    wave_merge_inverse_q4(v, row, spi, lane);
  else
    wave_merge_inverse(v, row, spi, lane);
  wave_lds_fence();
  wave_fft512<true, Q4>(v, tile, twl, tw, lane, xt);
  BT8_STAMP(9)
  const float inv_n = 1.0f / (float)(Q4 ? 256 : N);  // a power of two: x * (1/N) == x / N exactly
  cpx* ola = reinterpret_cast<cpx*>(sq);  // [frame][256]: second halves of frames 0..6
  // (barrier 3 above: every wave is done with the block gains that share the table)
  if constexpr (Q4) {
    // v[2 q] / v[2 q + 1] = samples 2 lane, 2 lane + 1 of the first / second half of stream q's frame
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const cpx t = v[2 * q + 1] * inv_n;
      if (wave < 7) ola[wave * 256 + 64 * q + lane] = t;
      v[2 * q + 1] = t;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x2 acc;
      if (wave == 0) {
        acc = tail[q];
      } else {
        const cpx pr = ola[(wave - 1) * 256 + 64 * q + lane];
        acc = f32x2{0.0f, 0.0f} + pr;
      }
      acc = acc + v[2 * q] * inv_n;
      *reinterpret_cast<f32x2*>(out + (g4 + q) * out_stride + HALFV * wave + 2 * lane) = acc;
    }
    if (wave == 7) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* dst = state + (g4 + q) * kStateFloats + kOffOutTail + 2 * lane;
        if constexpr (FLOW) st2_sc1(dst, f32x2{0.0f, 0.0f} + v[2 * q + 1]);
        else *reinterpret_cast<f32x2*>(dst) = f32x2{0.0f, 0.0f} + v[2 * q + 1];
      }
      if constexpr (FLOW) {  // publish the four stream-channels' macroblock: the tail stores drained first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < 4) __hip_atomic_store((bt_gu32*)(fa.seq + g4 + lane), flow_want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else {
  // v[j] = samples 2p, 2p + 1 of the frame, p = lane + 64 j
#pragma unroll
  for (int j = 4; j < 8; ++j) {
    const cpx t = v[j] * inv_n;
    if (wave < 7) ola[wave * 256 + lane + 64 * (j - 4)] = t;
    v[j] = t;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = lane + 64 * j;
    f32x2 acc;
    if (wave == 0) {
      acc = tail[j];
    } else {
      const cpx pr = ola[(wave - 1) * 256 + p];
      acc = f32x2{0.0f, 0.0f} + pr;
    }
    acc = acc + v[j] * inv_n;
    *reinterpret_cast<f32x2*>(y + HALF * wave + 2 * p) = acc;
  }
  if (wave == 7) {
#pragma unroll
    for (int j = 4; j < 8; ++j) {
      float* dst = st + kOffOutTail + 2 * (lane + 64 * (j - 4));
      if constexpr (FLOW) st2_sc1(dst, f32x2{0.0f, 0.0f} + v[j]);
      else *reinterpret_cast<f32x2*>(dst) = f32x2{0.0f, 0.0f} + v[j];
    }
    if constexpr (FLOW) {  // publish the stream-channel's macroblock: the tail stores drained first
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store((bt_gu32*)(fa.seq + blockIdx.x), flow_want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  }
  BT8_STAMP(10)
  BT8_WSTAMP(3)
}

// kiss_fftr / kiss_fftri seam for N = 1024 through the same wave routines: one wave per row.
__global__ __launch_bounds__(64) void bt_fftr8_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      int inverse, const BtTables* __restrict__ Tb) {
  __shared__ __align__(256) cpx row[ROW];
  __shared__ __align__(16) cpx twl[kTwLds];
  const int lane = threadIdx.x, r = blockIdx.x;
  const cpx* tw = reinterpret_cast<const cpx*>(Tb->tw1024_f);
  const cpx* sup = reinterpret_cast<const cpx*>(Tb->sup1024_f);
  for (int k = lane; k < kTwLds; k += 64) twl[k] = tw[k];
  wave_lds_fence();
  const XTerms xt = exchange_terms(Tb, lane, 0);
  cpx* tile = row;
  cpx v[8];
  if (!inverse) {
    const cpx* xin = reinterpret_cast<const cpx*>(src + (size_t)r * N);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = xin[lane + la_input(j)];
    wave_fft512<false>(v, tile, twl, tw, lane, xt);
#pragma unroll
    for (int j = 0; j < 8; ++j) row[lane + 64 * j] = v[j];
    wave_lds_fence();
    cpx spf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) spf[i] = sup[lane + 64 * i];
    wave_split_forward<false>(row, spf, lane, nullptr, 0.f);
    wave_lds_fence();
    cpx* o = reinterpret_cast<cpx*>(dst + (size_t)r * 2 * NB);
    for (int k = lane; k < NB; k += 64) o[k] = row[k];
  } else {
    const cpx* f = reinterpret_cast<const cpx*>(src + (size_t)r * 2 * NB);
    for (int k = lane; k < NB; k += 64) row[k] = f[k];
    wave_lds_fence();
    cpx spi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) spi[j] = sup[merge_sup_index(lane, j)];
    wave_merge_inverse(v, row, spi, lane);
    wave_lds_fence();
    wave_fft512<true>(v, tile, twl, tw, lane, xt);
    cpx* o = reinterpret_cast<cpx*>(dst + (size_t)r * N);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[lane + 64 * j] = v[j];
  }
}

}  // namespace

namespace aspbt {

hipError_t launch_bt_macroblock8(float* state, const BtTables* T, const float* in, float* out,
                                 int num_streams, int in_stride, int out_stride, hipStream_t s,
                                 unsigned long long* stamps) {
  const BtFlowArgs none = {nullptr, nullptr, 0u, 0, 1, 0u};
  hipLaunchKernelGGL((bt_macroblock8_kernel<false, false>), dim3(num_streams), dim3(512), 0, s, state, T, in, out,
                     in_stride, out_stride, stamps, none);
  return hipGetLastError();
}

hipError_t launch_bt_macroblock8_q4(float* state, const BtTables* T, const float* in, float* out, int groups,
                                    int in_stride, int out_stride, hipStream_t s) {
  const BtFlowArgs none = {nullptr, nullptr, 0u, 0, 1, 0u};
  hipLaunchKernelGGL((bt_macroblock8_kernel<true, false>), dim3(groups), dim3(512), 0, s, state, T, in, out, in_stride,
                     out_stride, (unsigned long long*)nullptr, none);
  return hipGetLastError();
}

// The hand-off build: `steps` consecutive macroblocks of every stream-channel in one launch (grid y = step); step j
// takes ring slot (slot0 + j) % ring of in / out (slots `per` floats apart, a stream-channel's macroblock `stride`
// floats from the next one's).  q4: the 256-sample window's four-stream-channels-per-workgroup form (num_streams % 4 == 0).
hipError_t launch_bt_macroblock8_flow(bool q4, float* state, const BtTables* T, const float* in, float* out, int num_streams,
                                      int stride, hipStream_t s, unsigned* seq, unsigned* abort_w, unsigned want, int steps,
                                      int slot0, int ring, size_t per) {
  const BtFlowArgs fa = {seq, abort_w, want, slot0, ring, (unsigned)per};
  if (q4)
    hipLaunchKernelGGL((bt_macroblock8_kernel<true, true>), dim3(num_streams / 4, steps), dim3(512), 0, s, state, T, in, out,
                       stride, stride, (unsigned long long*)nullptr, fa);
  else
    hipLaunchKernelGGL((bt_macroblock8_kernel<false, true>), dim3(num_streams, steps), dim3(512), 0, s, state, T, in, out,
                       stride, stride, (unsigned long long*)nullptr, fa);
  return hipGetLastError();
}

hipError_t launch_bt_fftr8(const float* src, float* dst, int count, int inverse, const BtTables* T,
                           hipStream_t s) {
  hipLaunchKernelGGL(bt_fftr8_kernel, dim3(count), dim3(64), 0, s, src, dst, inverse, T);
  return hipGetLastError();
}

}  // namespace aspbt
