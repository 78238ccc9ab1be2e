// bt_kernels8.hip -- the BlockThresholding macroblock kernel for N = 1024, one wave per STFT frame.
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

using namespace aspbt;

namespace {

struct cpx {
  float r, i;
};
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int N = 1024, NC = 512, HALF = 512, NB = 513, NCOL = 31;
constexpr int ROW = NB;   // complex slots per coefficient row
constexpr int SQS = 33;   // floats per row of the squared-real table [r * 16 + cc][column]

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ cpx cmul(cpx a, cpx b) {  // C_MUL
  cpx m;
  m.r = a.r * b.r - a.i * b.i;
  m.i = a.r * b.i + a.i * b.r;
  return m;
}
__device__ __forceinline__ cpx conj(cpx a) {
  a.i = -a.i;
  return a;
}

// ---------------------------------------------------------------- register layouts of the FFT stages
// A position p (9 bits) of the in-place kiss_fft work array lives in lane `lane`, register j of a
// layout: reg[b] / lane[b] name the position bit held by register-index bit b / lane bit b.
struct Lay {
  int reg[3];
  int lane[6];
};
constexpr Lay LA = {{0, 1, 2}, {7, 8, 5, 6, 3, 4}};  // after the loads: lane = k0 + 4 k1 + 16 k2 of n
constexpr Lay LB = {{8, 3, 4}, {0, 1, 2, 5, 6, 7}};  // radix-4 stage m = 8 (position bits 3, 4 in registers)
constexpr Lay LC = {{8, 5, 6}, {0, 1, 2, 3, 4, 7}};  // m = 32
constexpr Lay LD = {{6, 7, 8}, {0, 1, 2, 3, 4, 5}};  // m = 128; natural order: p = lane + 64 j
// XOR swizzles of the three exchanges (index bit b = parity(p & rows[b]) for b < 5, bits 5..8 kept):
// found by search so that the ds_write_b64 of the source layout (16-lane groups, 16 8-byte banks) and
// the ds_read_b64 of the destination layout (32-lane groups, 32 8-byte banks) are both conflict-free.
struct Swz {
  int rows[5];
};
constexpr Swz S1 = {{257, 322, 396, 40, 80}};
constexpr Swz S2 = {{129, 34, 4, 56, 16}};
constexpr Swz S3 = {{17, 66, 12, 264, 16}};

__host__ __device__ constexpr int pos_reg(const Lay& L, int j) {
  int p = 0;
  for (int b = 0; b < 3; ++b) p |= ((j >> b) & 1) << L.reg[b];
  return p;
}
__device__ __forceinline__ int pos_lane(const Lay& L, int lane) {
  int p = 0;
#pragma unroll
  for (int b = 0; b < 6; ++b) p |= ((lane >> b) & 1) << L.lane[b];
  return p;
}
__host__ __device__ constexpr int parity9(int x) {
  x ^= x >> 8;
  x ^= x >> 4;
  x ^= x >> 2;
  x ^= x >> 1;
  return x & 1;
}
__host__ __device__ constexpr int swz(const Swz& S, int p) {
  int r = p & ~31;
  for (int b = 0; b < 5; ++b) r |= parity9(p & S.rows[b]) << b;
  return r;
}

// One exchange through the wave's private 512-slot LDS row: registers of layout SRC -> layout DST.
// swz is linear over GF(2) and a position is the XOR of its lane part and its register part, so the
// slot is (lane term) ^ (compile-time register term).
struct XTerms {  // the six lane terms, computed once per kernel
  int w[3], r[3];
};
__device__ __forceinline__ XTerms exchange_terms(int lane) {
  XTerms t;
  t.w[0] = swz(S1, pos_lane(LA, lane));
  t.r[0] = swz(S1, pos_lane(LB, lane));
  t.w[1] = swz(S2, pos_lane(LB, lane));
  t.r[1] = swz(S2, pos_lane(LC, lane));
  t.w[2] = swz(S3, pos_lane(LC, lane));
  t.r[2] = swz(S3, pos_lane(LD, lane));
  return t;
}
template <int X>
__device__ __forceinline__ void exchange(cpx (&v)[8], cpx* row, const XTerms& xt) {
  constexpr Lay SRC = X == 1 ? LA : X == 2 ? LB : LC;
  constexpr Lay DST = X == 1 ? LB : X == 2 ? LC : LD;
  constexpr Swz S = X == 1 ? S1 : X == 2 ? S2 : S3;
  // laundered so that the 16 slot addresses are recomputed here (one XOR each) instead of being kept
  // alive -- and spilled -- between the forward and the inverse transform
  int wl = xt.w[X - 1], rl = xt.r[X - 1];
  asm volatile("" : "+v"(wl), "+v"(rl));
#pragma unroll
  for (int j = 0; j < 8; ++j) row[wl ^ swz(S, pos_reg(SRC, j))] = v[j];
  wave_lds_fence();
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = row[rl ^ swz(S, pos_reg(DST, j))];
  wave_lds_fence();
}

// kf_bfly2 (kiss_fft.c:21-42), m = 1
__device__ __forceinline__ void bfly2(cpx& f0, cpx& f1, cpx tw0) {
  const cpx t = cmul(f1, tw0);
  const cpx a = f0;
  f1.r = a.r - t.r;
  f1.i = a.i - t.i;
  f0.r = a.r + t.r;
  f0.i = a.i + t.i;
}
// kf_bfly4 (kiss_fft.c:44-90): a0..a3 = Fout[0], Fout[m], Fout[2m], Fout[3m]
template <bool INV>
__device__ __forceinline__ void bfly4(cpx& a0, cpx& a1, cpx& a2, cpx& a3, cpx t1, cpx t2, cpx t3) {
  const cpx s0 = cmul(a1, t1);
  const cpx s1 = cmul(a2, t2);
  const cpx s2 = cmul(a3, t3);
  cpx f0 = a0, s3, s4, s5;
  s5.r = f0.r - s1.r;
  s5.i = f0.i - s1.i;
  f0.r += s1.r;
  f0.i += s1.i;
  s3.r = s0.r + s2.r;
  s3.i = s0.i + s2.i;
  s4.r = s0.r - s2.r;
  s4.i = s0.i - s2.i;
  a2.r = f0.r - s3.r;
  a2.i = f0.i - s3.i;
  f0.r += s3.r;
  f0.i += s3.i;
  a0 = f0;
  if (INV) {
    a1.r = s5.r - s4.i;
    a1.i = s5.i + s4.r;
    a3.r = s5.r + s4.i;
    a3.i = s5.i - s4.r;
  } else {
    a1.r = s5.r + s4.i;
    a1.i = s5.i - s4.r;
    a3.r = s5.r - s4.i;
    a3.i = s5.i + s4.r;
  }
}

// Twiddles: the forward tables (the inverse ones are their conjugates bit for bit -- bt_api.hip checks
// that when it builds them) are staged in LDS once per workgroup (kTwLds entries of kiss_fft's table,
// all the stages below index, and the 256 super twiddles of kiss_fftr) and read just ahead of each stage.
constexpr int kTwLds = 384, kSupLds = 256;
template <bool INV>
__device__ __forceinline__ cpx tw_at(const cpx* twl, int idx) {
  const cpx t = twl[idx];
  return INV ? conj(t) : t;
}

// All stages of one 512-point complex FFT of kiss_fft (factors 4,4,4,4,2): in: layout LA, out: LD.
// tw: the global table (wave-uniform entries of the first two stages), twl: its LDS copy
template <bool INV>
__device__ __forceinline__ void wave_fft512(cpx (&v)[8], cpx* row, const cpx* twl,
                                            const cpx* __restrict__ tw, int lane, const XTerms& xt) {
  {  // radix-2 leaves (m = 1, twiddle 0) and the radix-4 stage with m = 2: wave-uniform twiddles
    cpx t0 = tw[0], t64 = tw[64], t128 = tw[128], t192 = tw[192];
    if (INV) t0 = conj(t0), t64 = conj(t64), t128 = conj(t128), t192 = conj(t192);
    bfly2(v[0], v[1], t0);
    bfly2(v[2], v[3], t0);
    bfly2(v[4], v[5], t0);
    bfly2(v[6], v[7], t0);
    bfly4<INV>(v[0], v[2], v[4], v[6], t0, t0, t0);
    bfly4<INV>(v[1], v[3], v[5], v[7], t64, t128, t192);
  }
  {
    const int kb = lane & 7;
    const cpx t1 = tw_at<INV>(twl, kb * 16), t2 = tw_at<INV>(twl, kb * 32), t3 = tw_at<INV>(twl, kb * 48);
    exchange<1>(v, row, xt);
#pragma unroll
    for (int c = 0; c < 2; ++c) bfly4<INV>(v[c], v[c + 2], v[c + 4], v[c + 6], t1, t2, t3);
  }
  {
    const int kc = lane & 31;
    const cpx t1 = tw_at<INV>(twl, kc * 4), t2 = tw_at<INV>(twl, kc * 8), t3 = tw_at<INV>(twl, kc * 12);
    exchange<2>(v, row, xt);
#pragma unroll
    for (int c = 0; c < 2; ++c) bfly4<INV>(v[c], v[c + 2], v[c + 4], v[c + 6], t1, t2, t3);
  }
  {
    cpx t[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 3; ++q) t[c][q] = tw_at<INV>(twl, (lane + 64 * c) * (q + 1));
    exchange<3>(v, row, xt);
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
__device__ __forceinline__ void wave_split_forward(cpx* row, const cpx* sup, int lane,
                                                   float* sq_fr, float norm) {
  cpx fp[4], fn[4], sp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 1 + lane + 64 * i;
    fp[i] = row[k];
    fn[i] = row[NC - k];
    sp[i] = sup[k - 1];
  }
  const cpx t0 = row[0];
  wave_lds_fence();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 1 + lane + 64 * i;
    cpx fpnk, f1k, f2k;
    fpnk.r = fn[i].r;
    fpnk.i = -fn[i].i;
    f1k.r = fp[i].r + fpnk.r;
    f1k.i = fp[i].i + fpnk.i;
    f2k.r = fp[i].r - fpnk.r;
    f2k.i = fp[i].i - fpnk.i;
    const cpx twv = cmul(f2k, sp[i]);
    cpx a, b;
    a.r = (f1k.r + twv.r) * 0.5f;
    a.i = (f1k.i + twv.i) * 0.5f;
    b.r = (f1k.r - twv.r) * 0.5f;
    b.i = (twv.i - f1k.i) * 0.5f;
    if (k != NC - k) row[k] = a;
    row[NC - k] = b;  // for k = NC/2 the second assignment is the one that stays
    if (SQ) {
      // bins 1 .. 16 NCOL belong to macro-columns: bin = 1 + 16 m + cc
      const int ka = k - 1, kb = NC - k - 1;
      if (k != NC - k) {
        const float va = a.r * norm;
        sq_fr[(ka & 15) * SQS + (ka >> 4)] = va * va;
      }
      if (kb < 16 * NCOL) {
        const float vb = b.r * norm;
        sq_fr[(kb & 15) * SQS + (kb >> 4)] = vb * vb;
      }
    }
  }
  if (lane == 0) {
    cpx z;
    z.r = t0.r + t0.i;
    z.i = 0.f;
    row[0] = z;
    z.r = t0.r - t0.i;
    row[NC] = z;
  }
}

// kiss_fftri pre-pass (kiss_fftr.c:137-157) from the row (natural order F[0..512]) straight into the
// registers of layout LA: register j wants T[n], n = lane + la_input(j).
__device__ __forceinline__ void wave_merge_inverse(cpx (&v)[8], const cpx* row, const cpx* sup,
                                                   int lane) {
  cpx u[8], w[8], sp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = lane + la_input(j);
    u[j] = row[n];
    w[j] = row[NC - n];
    const int k = (j & 1) ? NC - n : n;      // n >= 256: the (NC - k) side of pair k = NC - n
    sp[j] = conj(sup[k > 0 ? k - 1 : 0]);    // inverse super twiddles = conjugates of the forward ones
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const cpx fk = (j & 1) ? w[j] : u[j], fo = (j & 1) ? u[j] : w[j];  // F[k], F[NC - k]
    cpx fek, tmp;
    fek.r = fk.r + fo.r;
    fek.i = fk.i - fo.i;
    tmp.r = fk.r - fo.r;
    tmp.i = fk.i + fo.i;
    const cpx fok = cmul(tmp, sp[j]);
    cpx t;
    if (j & 1) {
      t.r = fek.r - fok.r;
      t.i = (fek.i - fok.i) * -1;
    } else {
      t.r = fek.r + fok.r;
      t.i = fek.i + fok.i;
    }
    if (j == 0 && lane == 0) {  // n = 0
      t.r = u[0].r + w[0].r;
      t.i = u[0].r - w[0].r;
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

// one term of the SURE sum (.c:391-398)
__device__ __forceinline__ float sure_term(float e, const BtSeg& sg) {
  const float q = fdiv_checked(sg.temp, e);
  return sg.size_blk + q * (float)(e > sg.thr) + (e - sg.two_size) * (float)(e <= sg.thr);
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
__device__ __forceinline__ void sure_seg(const float* sq, float* sure, const BtSeg sg, int lane) {
  constexpr int TT = 8 >> T, FF = 16 >> F, S = T + F, c = T * 5 + F;
  const int h = lane >> 5, m = lane & 31;
  if constexpr (S == 0) {
    const float e = block_sum<8, 16>(sq + m, 0, 0);
    float s = 0.0f;
    s += sure_term(e, sg);
    if (lane < NCOL) sure[m * 16 + c] = s;
  } else {
    constexpr int NTERM = 1 << (S - 1);
    constexpr int NI = T >= 1 ? (1 << (T - 1)) : 1;  // ii values per half
    constexpr int NJ = T >= 1 ? (1 << F) : (1 << (F - 1));  // jj values per half
    const float* col = sq + m + (T >= 1 ? h * (4 * 16 * SQS) : h * (8 * SQS));
    float t[NTERM];
#pragma unroll
    for (int il = 0; il < NI; ++il)
#pragma unroll
      for (int jl = 0; jl < NJ; ++jl)
        t[il * NJ + jl] = sure_term(block_sum<TT, FF>(col, TT * il, FF * jl), sg);
    float s0 = 0.0f;
#pragma unroll
    for (int q = 0; q < NTERM; ++q) s0 += t[q];
    float s1 = __shfl_xor(s0, 32);  // the first half's total, seen from the second half
#pragma unroll
    for (int q = 0; q < NTERM; ++q) s1 += t[q];
    if (h == 1 && m < NCOL) sure[m * 16 + c] = s1;
  }
}

#define BT8_STAMP(k) \
  if (stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime();

__global__ __launch_bounds__(512, 4) void bt_macroblock8_kernel(
    float* __restrict__ state, const BtTables* __restrict__ Tb, const float* __restrict__ in,
    float* __restrict__ out, int in_stride, int out_stride, unsigned long long* __restrict__ stamps) {
  __shared__ __align__(16) cpx coef[8 * ROW];      // [frame][bin]; a wave's row is also its exchange buffer
  __shared__ __align__(16) float sq[128 * SQS];    // squared normalised real parts; later block gains, OLA halves
  __shared__ float sure[32 * 16];                  // [macro-column][segmentation]
  __shared__ __align__(16) cpx twl[kTwLds];        // kiss_fft twiddles 0..383 (forward)
  __shared__ __align__(16) cpx supl[kSupLds];      // kiss_fftr super twiddles (forward)
  BT8_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the phase switches are scalar branches
  const int stream = blockIdx.x;
  float* st = state + (size_t)stream * kStateFloats;
  const float* x = in + (size_t)stream * in_stride;
  float* y = out + (size_t)stream * out_stride;
  const BtSize& P = Tb->s1024;
  const cpx* tw = reinterpret_cast<const cpx*>(Tb->tw1024_f);
  const cpx* sup = reinterpret_cast<const cpx*>(Tb->sup1024_f);
  cpx* row = coef + wave * ROW;
  const XTerms xt = exchange_terms(lane);

  // ---------------------------------------------------------------- phase A: STFT of frame `wave`
  // (blockThreshold_STFT, .c:273-282): frame t sees B[HALF t .. HALF t + N) of B = [inbuf tail | new samples]
  cpx v[8];
  {
    const f32x2* hann2 = reinterpret_cast<const f32x2*>(Tb->hann1024);
    f32x2 s[8], hw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = lane + la_input(j);  // complex input index: samples 2n, 2n + 1 of the frame
      const float* src = (wave == 0 && (j & 1) == 0) ? st + kOffInTail + 2 * n : x + HALF * (wave - 1) + 2 * n;
      s[j] = *reinterpret_cast<const f32x2*>(src);
      hw[j] = hann2[n];
    }
    if (tid < kTwLds) twl[tid] = tw[tid];
    if (tid < kSupLds) supl[tid] = sup[tid];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j].r = s[j].x * hw[j].x;
      v[j].i = s[j].y * hw[j].y;
    }
    __syncthreads();  // the twiddle tables are staged (every wave is waiting for its samples here anyway)
    BT8_STAMP(1)
    wave_fft512<false>(v, row, twl, tw, lane, xt);
  }
  BT8_STAMP(2)
#pragma unroll
  for (int j = 0; j < 8; ++j) row[lane + 64 * j] = v[j];
  wave_lds_fence();
  wave_split_forward<true>(row, supl, lane, sq + wave * 16 * SQS, P.norm);
  BT8_STAMP(3)
  __syncthreads();
  BT8_STAMP(4)
  // carry the last HALF input samples (wave 0 has read the old tail before the barrier)
  st[kOffInTail + tid] = x[7 * HALF + tid];

  // ---------------------------------------------------------------- phase B1: SURE (.c:354-401)
  switch (wave) {  // segmentations dealt by cost (terms per half-wave: 32, 16, 16, 8, ...)
    case 0: sure_seg<2, 4>(sq, sure, P.seg[2][4], lane); break;
    case 1: sure_seg<2, 3>(sq, sure, P.seg[2][3], lane); sure_seg<0, 1>(sq, sure, P.seg[0][1], lane); break;
    case 2: sure_seg<1, 4>(sq, sure, P.seg[1][4], lane); sure_seg<1, 0>(sq, sure, P.seg[1][0], lane); break;
    case 3: sure_seg<2, 2>(sq, sure, P.seg[2][2], lane); sure_seg<0, 0>(sq, sure, P.seg[0][0], lane); break;
    case 4: sure_seg<1, 3>(sq, sure, P.seg[1][3], lane); sure_seg<2, 1>(sq, sure, P.seg[2][1], lane); break;
    case 5: sure_seg<0, 4>(sq, sure, P.seg[0][4], lane); sure_seg<1, 2>(sq, sure, P.seg[1][2], lane); break;
    case 6: sure_seg<0, 3>(sq, sure, P.seg[0][3], lane); sure_seg<0, 2>(sq, sure, P.seg[0][2], lane);
            sure_seg<1, 1>(sq, sure, P.seg[1][1], lane); break;
    default: sure_seg<2, 0>(sq, sure, P.seg[2][0], lane); break;
  }
  BT8_STAMP(5)
  __syncthreads();
  BT8_STAMP(6)

  // ---------------------------------------------------------------- phase B2: attenuation + Wiener, in place
  {
    float* av = sq;  // [macro-column][64 block gains]; the squared-real table is dead
    const int m = tid >> 4, u = tid & 15;
    if (m < NCOL) {
      // argmin, first minimum wins (.c:404-416)
      float best = sure[m * 16];
      int bc = 0;
#pragma unroll
      for (int c = 1; c < 15; ++c) {
        const float s = sure[m * 16 + c];
        if (s < best) {
          best = s;
          bc = c;
        }
      }
      const int T = bc >= 10 ? 2 : bc >= 5 ? 1 : 0, F = bc - 5 * T, S = T + F;
      const int TT = 8 >> T, FF = 16 >> F, len = 128 >> S, nblk = 1 << S;
      const float a_const = P.seg[T][F].a_const;
      cpx* col = coef + 1 + 16 * m;
      // block powers of the chosen segmentation, each summed rows outer / columns inner (.c:421-454)
      for (int b = u; b < nblk; b += 16) {
        const int ii = b >> F, jj = b & ((1 << F) - 1);
        const cpx* blk = col + ii * TT * ROW + jj * FF;
        float power = 0.0f;
        for (int i0 = 0; i0 < len; i0 += 8) {
          cpx z[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int i = i0 + q < len ? i0 + q : 0;
            z[q] = blk[(i >> (4 - F)) * ROW + (i & (FF - 1))];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (i0 + q < len) power += z[q].r * z[q].r + z[q].i * z[q].i;
        }
        float a = (float)(1.0 - (double)fdiv_checked(a_const, power));
        a = a * (float)(a > 0);
        av[m * 64 + b] = a;
      }
      wave_lds_fence();  // a column's 16 lanes sit in one wave
      // thresholded coefficient -> empirical Wiener gain on the original one (.c:446-452, 469-486)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const cpx z = col[r * ROW + u];
        const float a = av[m * 64 + ((r >> (3 - T)) << F) + (u >> (4 - F))];
        const float tr = z.r * a, ti = z.i * a;
        float wn = tr * tr + ti * ti;
        const float den = wn + P.wiener_c;
        float g = fdiv_lean(wn, den);
        if (__builtin_expect(!(den <= 1e18f && (wn >= 1e-30f || wn == 0.0f)), 0)) g = wn / den;
        cpx o;
        o.r = z.r * g;
        o.i = z.i * g;
        col[r * ROW + u] = o;
      }
    } else {
      // DC column and the bins past the last whole macro-column (.c:501-506, 518-532); the Nyquist bin
      // (512) is thresholded by the reference but never reaches the Wiener step: it stays as it is
      const int colx = u == 0 ? 0 : 16 * NCOL + u;
      float sum = 0.0f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const cpx z = coef[t * ROW + colx];
        sum += z.r * z.r + z.i * z.i;
      }
      float a = 1 - P.dc_const / sum;
      if (a < 0) a = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const cpx z = coef[t * ROW + colx];
        const float tr = z.r * a, ti = z.i * a;
        float wn = tr * tr + ti * ti;
        wn = wn / (wn + P.wiener_c);
        cpx o;
        o.r = z.r * wn;
        o.i = z.i * wn;
        coef[t * ROW + colx] = o;
      }
    }
  }
  // the old output tail, for wave 0's overlap-add (requested ahead of the barrier)
  f32x2 tail[4];
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) tail[j] = *reinterpret_cast<const f32x2*>(st + kOffOutTail + 2 * (lane + 64 * j));
  }
  BT8_STAMP(7)
  __syncthreads();
  BT8_STAMP(8)

  // ---------------------------------------------------------------- phase C: inverse STFT + overlap-add
  // (blockThreshold_inverse_STFT, .c:284-300)
  wave_merge_inverse(v, row, supl, lane);
  wave_lds_fence();
  wave_fft512<true>(v, row, twl, tw, lane, xt);
  BT8_STAMP(9)
  const float inv_n = 1.0f / (float)N;  // N is a power of two: x * (1/N) == x / N exactly
  cpx* ola = reinterpret_cast<cpx*>(sq);  // [frame][256]: second halves (samples 512..1023) of frames 0..6
  // v[j] = samples 2p, 2p + 1 of the frame, p = lane + 64 j
  // (barrier 3 above: every wave is done with the block gains that share the table)
#pragma unroll
  for (int j = 4; j < 8; ++j) {
    cpx t;
    t.r = v[j].r * inv_n;
    t.i = v[j].i * inv_n;
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
      acc.x = 0.0f + pr.r;
      acc.y = 0.0f + pr.i;
    }
    acc.x += v[j].r * inv_n;
    acc.y += v[j].i * inv_n;
    *reinterpret_cast<f32x2*>(y + HALF * wave + 2 * p) = acc;
  }
  if (wave == 7) {
#pragma unroll
    for (int j = 4; j < 8; ++j) {
      f32x2 t;
      t.x = 0.0f + v[j].r;
      t.y = 0.0f + v[j].i;
      *reinterpret_cast<f32x2*>(st + kOffOutTail + 2 * (lane + 64 * (j - 4))) = t;
    }
  }
  BT8_STAMP(10)
}

// kiss_fftr / kiss_fftri seam for N = 1024 through the same wave routines: one wave per row.
__global__ __launch_bounds__(64) void bt_fftr8_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      int inverse, const BtTables* __restrict__ Tb) {
  __shared__ __align__(16) cpx row[ROW];
  __shared__ __align__(16) cpx twl[kTwLds];
  __shared__ __align__(16) cpx supl[kSupLds];
  const int lane = threadIdx.x, r = blockIdx.x;
  const cpx* tw = reinterpret_cast<const cpx*>(Tb->tw1024_f);
  const cpx* sup = reinterpret_cast<const cpx*>(Tb->sup1024_f);
  for (int k = lane; k < kTwLds; k += 64) twl[k] = tw[k];
  for (int k = lane; k < kSupLds; k += 64) supl[k] = sup[k];
  wave_lds_fence();
  const XTerms xt = exchange_terms(lane);
  cpx v[8];
  if (!inverse) {
    const cpx* xin = reinterpret_cast<const cpx*>(src + (size_t)r * N);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = xin[lane + la_input(j)];
    wave_fft512<false>(v, row, twl, tw, lane, xt);
#pragma unroll
    for (int j = 0; j < 8; ++j) row[lane + 64 * j] = v[j];
    wave_lds_fence();
    wave_split_forward<false>(row, supl, lane, nullptr, 0.f);
    wave_lds_fence();
    cpx* o = reinterpret_cast<cpx*>(dst + (size_t)r * 2 * NB);
    for (int k = lane; k < NB; k += 64) o[k] = row[k];
  } else {
    const cpx* f = reinterpret_cast<const cpx*>(src + (size_t)r * 2 * NB);
    for (int k = lane; k < NB; k += 64) row[k] = f[k];
    wave_lds_fence();
    wave_merge_inverse(v, row, supl, lane);
    wave_lds_fence();
    wave_fft512<true>(v, row, twl, tw, lane, xt);
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
  hipLaunchKernelGGL(bt_macroblock8_kernel, dim3(num_streams), dim3(512), 0, s, state, T, in, out,
                     in_stride, out_stride, stamps);
  return hipGetLastError();
}

hipError_t launch_bt_fftr8(const float* src, float* dst, int count, int inverse, const BtTables* T,
                           hipStream_t s) {
  hipLaunchKernelGGL(bt_fftr8_kernel, dim3(count), dim3(64), 0, s, src, dst, inverse, T);
  return hipGetLastError();
}

}  // namespace aspbt
