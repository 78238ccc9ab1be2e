// bt_kernels.hip -- gfx950 kernels of the batched time-frequency block-thresholding
// denoiser.  Replaces, for many independent stream-channels per launch,
//   blockThreshold_STFT / _core / _adaptive_block / blockTreshold_compute_thre /
//   blockThreshold_wiener / blockThreshold_inverse_STFT
//   (Denoise/BlockThresholding/src/audioDenoiseBlockTreshold.c:273-539) and
//   kiss_fftr / kiss_fftri (common/kiss_fft/kiss_fftr.c:67-159, kiss_fft.c:21-302).
//
// Mapping: one workgroup (N/2 threads) per stream-channel macroblock (8 hops of N/2
// samples).  The eight N/2-point complex FFTs of a macroblock advance together, one
// radix stage per barrier, in a 32 KB LDS tile; inputs are scattered into kiss_fft's
// decimation order while loading, so every stage is in place.  The butterflies and the
// real-FFT split perform kiss_fft's float operations in kiss_fft's order, and every
// block sum of the SURE search / Stein attenuation runs rows-outer / columns-inner in
// one thread as the reference does, so results are bit-identical to oracle/bt_oracle.c.
// The 8 x (N/2+1) coefficient and attenuated tiles live in LDS (65.7 KB at N = 1024, two
// workgroups per CU); HBM sees only samples in, samples out and the two N/2-sample tails.
//
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include "bt_layout.h"
#include "bt_sure.h"
#include "pk_f32.h"

using namespace aspbt;

namespace {

struct cpx {
  float r, i;
};

__device__ __forceinline__ cpx cmul(cpx a, cpx b) {  // C_MUL
  cpx m;
  m.r = a.r * b.r - a.i * b.i;
  m.i = a.r * b.i + a.i * b.r;
  return m;
}

// position of input index n after kiss_fft's recursive decimation (kf_work,
// kiss_fft.c:237-302) for factors 4,4,..,4,2: n = k0 + 4 k1 + 16 k2 + ... ,
// position = k0 * (NC/4) + k1 * (NC/16) + ... + k_last.
template <int NC>
__device__ __forceinline__ int kiss_position(int n) {
  int pos = 0, m = NC;
#pragma unroll
  for (int rem = NC; rem > 2; rem >>= 2) {
    m >>= 2;
    pos += (n & 3) * m;
    n >>= 2;
  }
  return pos + n;  // the radix-2 leaf digit
}

template <int NC, int M>
__device__ __forceinline__ void radix4_stage(cpx* work, const cpx* __restrict__ tw, int frames,
                                             bool inverse, int tid) {
  if (M >= NC) return;
  constexpr int fstride = (4 * M <= NC) ? NC / (4 * M) : 1;
  for (int w = tid; w < frames * (NC / 4); w += NC) {
    const int fr = w / (NC / 4), b = w % (NC / 4);
    const int g = b / M, k = b % M;
    cpx* F = work + fr * NC + g * 4 * M + k;
    const cpx s0 = cmul(F[M], tw[k * fstride]);
    const cpx s1 = cmul(F[2 * M], tw[k * fstride * 2]);
    const cpx s2 = cmul(F[3 * M], tw[k * fstride * 3]);
    cpx f0 = F[0], s3, s4, s5;
    s5.r = f0.r - s1.r;
    s5.i = f0.i - s1.i;
    f0.r += s1.r;
    f0.i += s1.i;
    s3.r = s0.r + s2.r;
    s3.i = s0.i + s2.i;
    s4.r = s0.r - s2.r;
    s4.i = s0.i - s2.i;
    F[2 * M].r = f0.r - s3.r;
    F[2 * M].i = f0.i - s3.i;
    f0.r += s3.r;
    f0.i += s3.i;
    F[0] = f0;
    if (inverse) {
      F[M].r = s5.r - s4.i;
      F[M].i = s5.i + s4.r;
      F[3 * M].r = s5.r + s4.i;
      F[3 * M].i = s5.i - s4.r;
    } else {
      F[M].r = s5.r + s4.i;
      F[M].i = s5.i - s4.r;
      F[3 * M].r = s5.r - s4.i;
      F[3 * M].i = s5.i + s4.r;
    }
  }
  __syncthreads();
}

// All radix stages of `frames` NC-point FFTs held in work[frame][NC] (kiss order input).
template <int NC>
__device__ __forceinline__ void kiss_stages(cpx* work, const cpx* __restrict__ tw, int frames,
                                            bool inverse, int tid) {
  // radix-2 leaves, m = 1, twiddle index 0 (kf_bfly2, kiss_fft.c:21-42)
  for (int w = tid; w < frames * (NC / 2); w += NC) {
    const int fr = w / (NC / 2), b = w % (NC / 2);
    cpx* F = work + fr * NC + 2 * b;
    const cpx t = cmul(F[1], tw[0]);
    const cpx a = F[0];
    F[1].r = a.r - t.r;
    F[1].i = a.i - t.i;
    F[0].r = a.r + t.r;
    F[0].i = a.i + t.i;
  }
  __syncthreads();
  // radix-4 stages, m = 2, 8, 32, ... (kf_bfly4, kiss_fft.c:44-90); strides are compile-time
  radix4_stage<NC, 2>(work, tw, frames, inverse, tid);
  if (NC > 8) radix4_stage<NC, 8>(work, tw, frames, inverse, tid);
  if (NC > 32) radix4_stage<NC, 32>(work, tw, frames, inverse, tid);
  if (NC > 128) radix4_stage<NC, 128>(work, tw, frames, inverse, tid);
}

// kiss_fftr post-pass (kiss_fftr.c:92-120): work[fr][NC] -> freq[fr][NC+1]
template <int NC>
__device__ __forceinline__ void real_split_forward(const cpx* work, cpx* freq,
                                                   const cpx* __restrict__ sup, int frames,
                                                   int tid) {
  for (int w = tid; w < frames * (NC / 2 + 1); w += NC) {
    const int fr = w / (NC / 2 + 1), k = w % (NC / 2 + 1);
    const cpx* T = work + fr * NC;
    cpx* Fq = freq + fr * (NC + 1);
    if (k == 0) {
      const float tdr = T[0].r, tdi = T[0].i;
      Fq[0].r = tdr + tdi;
      Fq[0].i = 0.f;
      Fq[NC].r = tdr - tdi;
      Fq[NC].i = 0.f;
    } else {
      const cpx fpk = T[k];
      cpx fpnk, f1k, f2k;
      fpnk.r = T[NC - k].r;
      fpnk.i = -T[NC - k].i;
      f1k.r = fpk.r + fpnk.r;
      f1k.i = fpk.i + fpnk.i;
      f2k.r = fpk.r - fpnk.r;
      f2k.i = fpk.i - fpnk.i;
      const cpx twv = cmul(f2k, sup[k - 1]);
      const float a_r = (f1k.r + twv.r) * 0.5f, a_i = (f1k.i + twv.i) * 0.5f;
      const float b_r = (f1k.r - twv.r) * 0.5f, b_i = (twv.i - f1k.i) * 0.5f;
      if (k != NC - k) {
        Fq[k].r = a_r;
        Fq[k].i = a_i;
      }
      Fq[NC - k].r = b_r;  // for k = NC/2 the second assignment is the one that stays
      Fq[NC - k].i = b_i;
    }
  }
}

// kiss_fftri pre-pass (kiss_fftr.c:137-157): freq[fr][NC+1] -> work[fr][kiss order]
template <int NC>
__device__ __forceinline__ void real_merge_inverse(const cpx* freq, cpx* work,
                                                   const cpx* __restrict__ sup, int frames,
                                                   int tid) {
  for (int w = tid; w < frames * (NC / 2 + 1); w += NC) {
    const int fr = w / (NC / 2 + 1), k = w % (NC / 2 + 1);
    const cpx* Fq = freq + fr * (NC + 1);
    cpx* T = work + fr * NC;
    if (k == 0) {
      cpx v;
      v.r = Fq[0].r + Fq[NC].r;
      v.i = Fq[0].r - Fq[NC].r;
      T[kiss_position<NC>(0)] = v;
    } else {
      const cpx fk = Fq[k];
      cpx fnkc, fek, tmp;
      fnkc.r = Fq[NC - k].r;
      fnkc.i = -Fq[NC - k].i;
      fek.r = fk.r + fnkc.r;
      fek.i = fk.i + fnkc.i;
      tmp.r = fk.r - fnkc.r;
      tmp.i = fk.i - fnkc.i;
      const cpx fok = cmul(tmp, sup[k - 1]);
      cpx a, b;
      a.r = fek.r + fok.r;
      a.i = fek.i + fok.i;
      b.r = fek.r - fok.r;
      b.i = (fek.i - fok.i) * -1;
      if (k != NC - k) T[kiss_position<NC>(k)] = a;
      T[kiss_position<NC>(NC - k)] = b;
    }
  }
}

// Sequential (rows outer, columns inner) sum of one TT x FF block of a [8][16] tile stored
// with element stride ES and row stride 16 * ES; fully unrolled so the LDS reads pipeline
// while the adds keep the reference's order (energy_real_STFT / power_STFT, .c:303-338).
template <int TT, int FF, int ES>
__device__ __forceinline__ float block_sum(const float* tile, int r0, int c0) {
  float acc = 0.0f;
#pragma unroll
  for (int r = 0; r < TT; ++r)
#pragma unroll
    for (int c = 0; c < FF; ++c) acc += tile[((r0 + r) * 16 + (c0 + c)) * ES];
  return acc;
}

// SURE of segmentation (T, F) for one macro-column (.c:378-400).
template <int T, int F, int ES>
__device__ __forceinline__ float sure_of(const float* tile, const BtSeg sg) {
  constexpr int TT = 8 >> T, FF = 16 >> F;
  float SURE_real = 0.0f;
  for (int ii = 0; ii < (1 << T); ii++)
    for (int jj = 0; jj < (1 << F); jj++) {
      const float energy_real = block_sum<TT, FF, ES>(tile, TT * ii, FF * jj);
      SURE_real += sg.size_blk + sg.temp / energy_real * (float)(energy_real > sg.thr) +
                   (energy_real - sg.two_size) * (float)(energy_real <= sg.thr);
    }
  return SURE_real;
}

template <int ES>
__device__ __forceinline__ float sure_dispatch(int c, const float* tile, const BtSeg sg) {
  switch (c) {
    case 0: return sure_of<0, 0, ES>(tile, sg);
    case 1: return sure_of<0, 1, ES>(tile, sg);
    case 2: return sure_of<0, 2, ES>(tile, sg);
    case 3: return sure_of<0, 3, ES>(tile, sg);
    case 4: return sure_of<0, 4, ES>(tile, sg);
    case 5: return sure_of<1, 0, ES>(tile, sg);
    case 6: return sure_of<1, 1, ES>(tile, sg);
    case 7: return sure_of<1, 2, ES>(tile, sg);
    case 8: return sure_of<1, 3, ES>(tile, sg);
    case 9: return sure_of<1, 4, ES>(tile, sg);
    case 10: return sure_of<2, 0, ES>(tile, sg);
    case 11: return sure_of<2, 1, ES>(tile, sg);
    case 12: return sure_of<2, 2, ES>(tile, sg);
    case 13: return sure_of<2, 3, ES>(tile, sg);
    default: return sure_of<2, 4, ES>(tile, sg);
  }
}

template <int ES>
__device__ __forceinline__ float block_sum_dispatch(int c, const float* tile, int ii, int jj) {
#define BT_CASE(k, T, F) \
  case k: return block_sum<(8 >> T), (16 >> F), ES>(tile, (8 >> T) * ii, (16 >> F) * jj);
  switch (c) {
    BT_CASE(0, 0, 0) BT_CASE(1, 0, 1) BT_CASE(2, 0, 2) BT_CASE(3, 0, 3) BT_CASE(4, 0, 4)
    BT_CASE(5, 1, 0) BT_CASE(6, 1, 1) BT_CASE(7, 1, 2) BT_CASE(8, 1, 3) BT_CASE(9, 1, 4)
    BT_CASE(10, 2, 0) BT_CASE(11, 2, 1) BT_CASE(12, 2, 2) BT_CASE(13, 2, 3)
    default: return block_sum<2, 1, ES>(tile, 2 * ii, jj);
  }
#undef BT_CASE
}

// --------------------------------------------------------------------------
// One macroblock (frames = 8, threshold = 1) or one flush (frames < 8, threshold = 0)
// of every stream.
template <int N>
__global__ __launch_bounds__(N / 2) void bt_macroblock_kernel(
    float* __restrict__ state, const BtTables* __restrict__ Tb, const float* __restrict__ in,
    float* __restrict__ out, int frames, int threshold, int in_stride, int out_stride,
    unsigned long long* __restrict__ stamps) {
  constexpr int NC = N / 2, HALF = N / 2, NB = N / 2 + 1;
  // diagnostic phase stamps (never enabled by the product entry points): workgroup 0, thread 0
#define BT_STAMP(k)                                                                   \
  if (stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime();
  BT_STAMP(0)
  constexpr int NCOL = (N - 1) / 2 / 16;  // macro-columns of 16 bins (.c:493)
  extern __shared__ __align__(16) unsigned char smem[];
  cpx* coef = reinterpret_cast<cpx*>(smem);  // [8][NB]
  cpx* thre = coef + 8 * NB;                 // [8][NB]; FFT work tile aliases it
  cpx* work = thre;                          // [8][NC]
  float* sure = reinterpret_cast<float*>(thre + 8 * NB);  // [NCOL][15]

  const int tid = threadIdx.x;
  const int stream = blockIdx.x;
  float* st = state + (size_t)stream * kStateFloats;
  const float* x = in + (size_t)stream * in_stride;
  float* y = out + (size_t)stream * out_stride;
  const BtSize& P = N == 1024 ? Tb->s1024 : Tb->s256;
  const float* hann = N == 1024 ? Tb->hann1024 : Tb->hann256;
  const cpx* tw_f = reinterpret_cast<const cpx*>(N == 1024 ? Tb->tw1024_f : Tb->tw256_f);
  const cpx* tw_i = reinterpret_cast<const cpx*>(N == 1024 ? Tb->tw1024_i : Tb->tw256_i);
  const cpx* sup_f = reinterpret_cast<const cpx*>(N == 1024 ? Tb->sup1024_f : Tb->sup256_f);
  const cpx* sup_i = reinterpret_cast<const cpx*>(N == 1024 ? Tb->sup1024_i : Tb->sup256_i);

  // ---- STFT (blockThreshold_STFT, .c:273-282): frame t sees B[HALF t .. HALF t + N) of
  // B = [inbuf tail | new samples]; windowed pairs go straight to kiss order.
  const int total = frames * HALF;  // new samples this call
  for (int w = tid; w < frames * NC; w += NC) {
    const int fr = w / NC, n = w % NC;
    const int p0 = HALF * fr + 2 * n;  // position in B
    const float b0 = p0 < HALF ? st[kOffInTail + p0] : x[p0 - HALF];
    const float b1 = p0 + 1 < HALF ? st[kOffInTail + p0 + 1] : x[p0 + 1 - HALF];
    cpx z;
    z.r = b0 * hann[2 * n];
    z.i = b1 * hann[2 * n + 1];
    work[fr * NC + kiss_position<NC>(n)] = z;
  }
  __syncthreads();
  BT_STAMP(1)
  // carry the last HALF input samples (after every thread has read the old tail).  A flush
  // (threshold == 0) leaves the input history alone: the reference transforms a hop when it is
  // fed (.c:550-559) and its flush only inverse-transforms the cached spectra (.c:618-672), so the
  // pending hops -- which stay pending, and are fed again when their macroblock completes -- must
  // see the same history every time
  if (threshold)
    for (int i = tid; i < HALF; i += NC) {
      const int p = total + i;  // position in B of the new tail
      st[kOffInTail + i] = p < HALF ? st[kOffInTail + p] : x[p - HALF];
    }
  kiss_stages<NC>(work, tw_f, frames, false, tid);
  BT_STAMP(2)
  real_split_forward<NC>(work, coef, sup_f, frames, tid);
  __syncthreads();
  BT_STAMP(3)

  if (threshold) {
    // ---- squared normalised real parts, laid out [row * 16 + bin][column] so that the
    // SURE pass (lanes = macro-columns) reads consecutive LDS words (.c:365-375, 322-338)
    constexpr int SQW = NCOL <= 8 ? 8 : 32;      // columns per table row (7 or 31 used)
    static_assert(128 * SQW * 4 <= 8 * NB * 8, "squared-real table must fit the thre tile");
    float* sq = reinterpret_cast<float*>(thre);  // 128 x SQW floats, dead before thre is written
    for (int w = tid; w < 128 * SQW; w += NC) {
      const int m = w % SQW, e = w / SQW;  // e = r * 16 + cc
      float v2 = 0.0f;
      if (m < NCOL) {
        const float v = coef[(e >> 4) * NB + 1 + m * 16 + (e & 15)].r * P.norm;
        v2 = v * v;
      }
      sq[w] = v2;
    }
    __syncthreads();
    BT_STAMP(4)
    // ---- SURE of the 15 dyadic segmentations of every macro-column (.c:354-401).  A wave
    // takes whole segmentations (block shape is a compile-time constant inside the switch, so
    // the block sums unroll and their LDS reads pipeline), lane = macro-column; every sum runs
    // in the reference's order: blocks ii-major / jj-minor, rows outer, columns inner.
    {
      const int wave = tid >> 6, lane = tid & 63, nwaves = NC / 64;
      for (int c = wave; c < 15; c += nwaves) {
        const float v = sure_dispatch<SQW>(c, sq + (lane % SQW), P.seg[c / 5][c % 5]);
        if (lane < NCOL) sure[lane * 15 + c] = v;
      }
    }
    __syncthreads();
    BT_STAMP(5)
    // ---- DC column and the bins past the last whole macro-column (.c:501-506, 518-532)
    for (int w = tid; w < 1 + (NB - (1 + NCOL * 16)); w += NC) {
      const int col = w == 0 ? 0 : (1 + NCOL * 16) + (w - 1);
      float sum = 0.0f;
      for (int t = 0; t < 8; ++t) {
        const float r = coef[t * NB + col].r, i = coef[t * NB + col].i;
        sum += r * r + i * i;
      }
      float a = 1 - P.dc_const / sum;
      if (a < 0) a = 0;
      for (int t = 0; t < 8; ++t) {
        thre[t * NB + col].r = coef[t * NB + col].r * a;
        thre[t * NB + col].i = coef[t * NB + col].i * a;
      }
    }
    // ---- argmin (first wins, .c:404-416) + Stein attenuation of the chosen blocks
    // (.c:421-454).  One wave per macro-column: the 128 squared magnitudes go to a per-wave
    // LDS tile in parallel, the <= 64 block powers are summed from it one block per lane (in
    // the reference's order), then all lanes scale their two coefficients.
    {
      const int wave = tid >> 6, lane = tid & 63, nwaves = NC / 64;
      float* pw = sure + NCOL * 15 + 16 + wave * (128 + 64);  // [128] powers + [64] gains
      float* av = pw + 128;
      for (int m = wave; m < NCOL; m += nwaves) {
        const int base = 1 + m * 16;
        float best = sure[m * 15];
        int bc = 0;
        for (int c = 1; c < 15; ++c)
          if (sure[m * 15 + c] < best) {
            best = sure[m * 15 + c];
            bc = c;
          }
        const int T = bc / 5, F = bc % 5;
        cpx ce[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = lane + 64 * h;  // e = r * 16 + cc
          ce[h] = coef[(e >> 4) * NB + base + (e & 15)];
          pw[e] = ce[h].r * ce[h].r + ce[h].i * ce[h].i;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < (1 << (T + F))) {
          const float power = block_sum_dispatch<1>(bc, pw, lane >> F, lane & ((1 << F) - 1));
          float a = (float)(1.0 - (double)(P.seg[T][F].a_const / power));
          av[lane] = a * (float)(a > 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = lane + 64 * h, r = e >> 4, cc = e & 15;
          const float a = av[((r >> (3 - T)) << F) + (cc >> (4 - F))];
          thre[r * NB + base + cc].r = ce[h].r * a;
          thre[r * NB + base + cc].i = ce[h].i * a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
    BT_STAMP(6)
    // ---- empirical Wiener on bins 0 .. N/2-1, Nyquist untouched (.c:469-486)
    for (int w = tid; w < 8 * NC; w += NC) {
      const int t = w / NC, f = w % NC;
      const float r = thre[t * NB + f].r, i = thre[t * NB + f].i;
      float wiener = r * r + i * i;
      wiener = wiener / (wiener + P.wiener_c);
      coef[t * NB + f].r *= wiener;
      coef[t * NB + f].i *= wiener;
    }
    __syncthreads();
  }

  BT_STAMP(7)
  // ---- inverse STFT + overlap-add (blockThreshold_inverse_STFT, .c:284-300)
  real_merge_inverse<NC>(coef, work, sup_i, frames, tid);
  __syncthreads();
  BT_STAMP(8)
  kiss_stages<NC>(work, tw_i, frames, true, tid);
  BT_STAMP(9)
  const float* td = reinterpret_cast<const float*>(work);  // frame fr sample j at fr*N + j
  const float inv_n = 1.0f / (float)N;  // N is a power of two: x * (1/N) == x / N exactly
  for (int q = tid; q < total + HALF; q += NC) {
    const int t2 = q / HALF, t1 = t2 - 1;
    float v = q < HALF ? st[kOffOutTail + q] : 0.0f;
    if (t1 >= 0 && t1 < frames) v += td[t1 * N + (q - HALF * t1)] * inv_n;
    if (t2 < frames) v += td[t2 * N + (q - HALF * t2)] * inv_n;
    if (q < total)
      y[q] = v;
    else
      sure[q - total] = v;  // stage the new tail: other threads still read the old one
  }
  __syncthreads();
  for (int i = tid; i < HALF; i += NC)
    st[kOffOutTail + i] = threshold ? sure[i] : 0.0f;  // a flush leaves the tail cleared (.c:656-672)
  BT_STAMP(10)
#undef BT_STAMP
}

// --------------------------------------------------------------------------
// Any even window of 4 .. 2048 samples (20 ms at 16 kHz = 320, 50 ms = 800, 10 / 40 ms at 48 kHz = 480 / 1920, ...):
// the same phases with run-time sizes and kiss_fft's mixed-radix plan -- radix 4, 2, 3, 5 and the generic
// butterfly for larger primes (kiss_fft.c:21-235).  One butterfly per thread and stage, the reference's
// float operations in the reference's order (butterflies of a stage are independent, so their order
// across threads does not matter).  A plain path: 256 threads per macroblock, a barrier per stage.
constexpr int kAnyThreads = 256;   // the seam kernel's workgroup; the macroblock kernel runs 256 or 512 threads

// Complex arithmetic as packed f32 (pk_f32.h: a complex value is one register pair; every half is the IEEE
// operation of the reference's scalar spelling).  F / tw are the LDS arrays seen as pairs.
typedef asppk::f32x2 pcx;
__device__ __forceinline__ pcx* as_pcx(cpx* p) { return reinterpret_cast<pcx*>(p); }
__device__ __forceinline__ const pcx* as_pcx(const cpx* p) { return reinterpret_cast<const pcx*>(p); }

__device__ __forceinline__ void any_bfly2(cpx* F_, int m, const cpx* tw_, int fstride, int u) {
  pcx* F = as_pcx(F_);
  const pcx* tw = as_pcx(tw_);
  const pcx t = asppk::cmul<false>(F[m], tw[u * fstride]);
  const pcx a = F[0];
  F[m] = a - t;
  F[0] = a + t;
}

__device__ __forceinline__ void any_bfly3(cpx* F_, int m, const cpx* tw_, int fstride, int u) {  // kiss_fft.c:92-136
  pcx* F = as_pcx(F_);
  const pcx* tw = as_pcx(tw_);
  const int m2 = 2 * m;
  const pcx epi3 = tw[fstride * m];
  const pcx s1 = asppk::cmul<false>(F[m], tw[u * fstride]), s2 = asppk::cmul<false>(F[m2], tw[2 * u * fstride]);
  const pcx s3 = s1 + s2;
  pcx s0 = s1 - s2;
  pcx f0 = F[0];
  const pcx f1 = f0 - s3 * .5f;  // HALF_OF
  s0 = s0 * epi3.y;              // C_MULBYSCALAR
  f0 = f0 + s3;
  F[0] = f0;
  F[m2] = asppk::add_swap_sub_hi(f1, s0);  // {f1.r + s0.i, f1.i - s0.r}
  F[m] = asppk::add_swap_sub_lo(f1, s0);   // {f1.r - s0.i, f1.i + s0.r}
}

__device__ __forceinline__ void any_bfly4(cpx* F_, int M, const cpx* tw_, int fstride, int k, bool inverse) {
  pcx* F = as_pcx(F_);
  const pcx* tw = as_pcx(tw_);
  const pcx s0 = asppk::cmul<false>(F[M], tw[k * fstride]);
  const pcx s1 = asppk::cmul<false>(F[2 * M], tw[k * fstride * 2]);
  const pcx s2 = asppk::cmul<false>(F[3 * M], tw[k * fstride * 3]);
  pcx f0 = F[0];
  const pcx s5 = f0 - s1;
  f0 = f0 + s1;
  const pcx s3 = s0 + s2;
  const pcx s4 = s0 - s2;
  F[2 * M] = f0 - s3;
  F[0] = f0 + s3;
  const pcx hi = asppk::add_swap_sub_hi(s5, s4);  // {s5.r + s4.i, s5.i - s4.r}
  const pcx lo = asppk::add_swap_sub_lo(s5, s4);  // {s5.r - s4.i, s5.i + s4.r}
  F[M] = inverse ? lo : hi;
  F[3 * M] = inverse ? hi : lo;
}

__device__ __forceinline__ void any_bfly5(cpx* F_, int m, const cpx* tw_, int fstride, int u) {  // kiss_fft.c:138-197
  pcx* F = as_pcx(F_);
  const pcx* tw = as_pcx(tw_);
  const pcx ya = tw[fstride * m], yb = tw[fstride * 2 * m];
  const pcx s0 = F[0];
  const pcx s1 = asppk::cmul<false>(F[m], tw[u * fstride]), s2 = asppk::cmul<false>(F[2 * m], tw[2 * u * fstride]);
  const pcx s3 = asppk::cmul<false>(F[3 * m], tw[3 * u * fstride]), s4 = asppk::cmul<false>(F[4 * m], tw[4 * u * fstride]);
  const pcx s7 = s1 + s4, s10 = s1 - s4, s8 = s2 + s3, s9 = s2 - s3;
  F[0] = s0 + (s7 + s8);
  const pcx s5 = s0 + s7 * ya.x + s8 * yb.x;
  // scratch[6] = {s10.i ya.i + s9.i yb.i, -(s10.r ya.i) - s9.r yb.i} = {w.x, -w.y}: (-a) - b == -(a + b) bit for bit
  const pcx w = s10.yx * ya.y + s9.yx * yb.y;
  F[m] = asppk::add_sub_lo(s5, w);      // scratch[5] - scratch[6]
  F[4 * m] = asppk::add_sub_hi(s5, w);  // scratch[5] + scratch[6]
  const pcx s11 = s0 + s7 * yb.x + s8 * ya.x;
  // scratch[12] = {-(s10.i yb.i) + s9.i ya.i, s10.r yb.i - s9.r ya.i} = {-d.x, d.y}: (-a) + b == -(a - b) bit for bit
  const pcx d = s10.yx * yb.y - s9.yx * ya.y;
  F[2 * m] = asppk::add_sub_lo(s11, d);  // scratch[11] + scratch[12]
  F[3 * m] = asppk::add_sub_hi(s11, d);  // scratch[11] - scratch[12]
}

// kf_bfly_generic (kiss_fft.c:199-235), the unit u of the sub-transform starting at `base` (index into
// the frame's array; the twiddle index runs on the position inside the sub-transform)
__device__ __forceinline__ void any_bfly_generic(cpx* F_, int m, int p, const cpx* tw_, int fstride, int u, int norig) {
  pcx* F = as_pcx(F_);
  const pcx* tw = as_pcx(tw_);
  pcx scratch[kAnyMaxRadix];
  int k = u;
  for (int q1 = 0; q1 < p; ++q1, k += m) scratch[q1] = F[k];
  k = u;
  for (int q1 = 0; q1 < p; ++q1, k += m) {
    int twidx = 0;
    pcx acc = scratch[0];
    for (int q = 1; q < p; ++q) {
      twidx += fstride * k;
      if (twidx >= norig) twidx -= norig;
      acc = acc + asppk::cmul<false>(scratch[q], tw[twidx]);
    }
    F[k] = acc;
  }
}

// w / d for 0 <= w < 2^16 and 1 <= d <= 2^11 through one float multiply (rd = 1.0f / d): (w + 0.5) / d is at
// least 0.5 / d away from every integer, far more than the rounding of the product, so the truncation is exact.
// (The loops below run a handful of butterflies per thread: four integer divisions each were most of their cost.)
__device__ __forceinline__ int fast_div(int w, float rd) { return (int)(((float)w + 0.5f) * rd); }

// every stage of `frames` nc-point transforms in work[frame][nc] (inputs already in kiss order)
// GENERIC: the plan has a radix other than 2, 3, 4, 5 (the generic butterfly's scratch lives in private memory:
// kernels for the usual windows are built without it)
template <int THREADS, bool GENERIC>
__device__ __forceinline__ void any_stages(cpx* work, const cpx* __restrict__ tw, const BtAnyTables& A, int frames,
                                           bool inverse, int tid) {
  const int nc = A.nc;
  for (int s = A.nfac - 1; s >= 0; --s) {
    const int p = A.fac[2 * s], m = A.fac[2 * s + 1];
    const int span = p * m, fstride = nc / span, units = nc / p;  // butterflies per frame
    const float r_units = 1.0f / (float)units, r_m = 1.0f / (float)m;
    for (int w = tid; w < frames * units; w += THREADS) {
      const int fr = fast_div(w, r_units), b = w - fr * units;
      const int g = fast_div(b, r_m), u = b - g * m;
      cpx* F = work + fr * nc + g * span;
      switch (p) {
        case 2: any_bfly2(F + u, m, tw, fstride, u); break;
        case 3: any_bfly3(F + u, m, tw, fstride, u); break;
        case 4: any_bfly4(F + u, m, tw, fstride, u, inverse); break;
        case 5: any_bfly5(F + u, m, tw, fstride, u); break;
        default:
          if constexpr (GENERIC) any_bfly_generic(F, m, p, tw, fstride, u, nc);
          break;
      }
    }
    __syncthreads();
  }
}

// SQW: columns of the squared-real table (32 for windows of up to 1024 samples: at most 31 macro-columns; else 64)
template <int THREADS, int SQW, bool GENERIC>
__global__ __launch_bounds__(THREADS) void bt_macroblock_any_kernel(
    float* __restrict__ state, const BtAnyTables* __restrict__ Ap, const float* __restrict__ in, float* __restrict__ out, int frames,
    int threshold, int in_stride, int out_stride, unsigned long long* __restrict__ stamps) {
  const BtAnyTables& A = *Ap;  // the plan stays in memory (scalar loads where it is read)
  // diagnostic phase stamps (never enabled by the product entry points): workgroup 0, thread 0
#define BTA_STAMP(k) \
  if (stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime();
  const int N = A.n, NC = A.nc, HALF = A.nc, NB = A.nc + 1, NCOL = A.ncol;
  extern __shared__ __align__(16) unsigned char smem[];
  cpx* coef = reinterpret_cast<cpx*>(smem);                 // [8][NB]
  cpx* thre = coef + 8 * NB;                                // [8][NB]; the FFT work tile aliases it
  cpx* work = thre;                                         // [8][NC]
  // [128][SQW] squared real parts: inside the thre tile when that is large enough (it is dead until the
  // SURE values are out), else behind it (launch_bt_macroblock_any sizes the allocation the same way)
  const bool sq_in_thre = (size_t)8 * NB * sizeof(cpx) >= (size_t)128 * SQW * sizeof(float);
  cpx* twl = thre + 8 * NB;  // [NC] twiddles of the running direction (every butterfly reads three or four of them)
  float* after = reinterpret_cast<float*>(twl + NC);
  float* sq = sq_in_thre ? reinterpret_cast<float*>(thre) : after;
  float* sure = sq_in_thre ? after : after + 128 * SQW;  // [NCOL][15], then 16 spare, then 4 x (128 + 64)
  float* stage = sure + NCOL * 15 + 16 + (THREADS / 64) * (128 + 64);  // [HALF] new output tail
  const int tid = threadIdx.x;
  const int stream = blockIdx.x;
  float* st = state + (size_t)stream * kAnyStateFloats;
  const float* x = in + (size_t)stream * in_stride;
  float* y = out + (size_t)stream * out_stride;
  const BtSize& P = A.P;
  const cpx* tw_f = reinterpret_cast<const cpx*>(A.tw_f);
  const cpx* tw_i = reinterpret_cast<const cpx*>(A.tw_i);
  const cpx* sup_f = reinterpret_cast<const cpx*>(A.sup_f);
  const cpx* sup_i = reinterpret_cast<const cpx*>(A.sup_i);

  BTA_STAMP(0)
  for (int i = tid; i < NC; i += THREADS) twl[i] = tw_f[i];
  // ---- STFT (blockThreshold_STFT, .c:273-282)
  const int total = frames * HALF;
  const float r_nc = 1.0f / (float)NC, r_nh = 1.0f / (float)(NC / 2 + 1);
  for (int w = tid; w < frames * NC; w += THREADS) {
    const int fr = fast_div(w, r_nc), n = w - fr * NC;
    const int p0 = HALF * fr + 2 * n;
    const float b0 = p0 < HALF ? st[kOffInTail + p0] : x[p0 - HALF];
    const float b1 = p0 + 1 < HALF ? st[kOffInTail + p0 + 1] : x[p0 + 1 - HALF];
    cpx z;
    z.r = b0 * A.hann[2 * n];
    z.i = b1 * A.hann[2 * n + 1];
    work[fr * NC + A.perm[n]] = z;
  }
  __syncthreads();
  if (threshold)  // a flush leaves the input history alone (see bt_macroblock_kernel)
    for (int i = tid; i < HALF; i += THREADS) {
      const int p = total + i;
      st[kOffInTail + i] = p < HALF ? st[kOffInTail + p] : x[p - HALF];
    }
  BTA_STAMP(1)
  any_stages<THREADS, GENERIC>(work, twl, A, frames, false, tid);
  BTA_STAMP(2)
  // kiss_fftr post-pass (kiss_fftr.c:92-120)
  for (int w = tid; w < frames * (NC / 2 + 1); w += THREADS) {
    const int fr = fast_div(w, r_nh), k = w - fr * (NC / 2 + 1);
    const cpx* T = work + fr * NC;
    cpx* Fq = coef + fr * NB;
    if (k == 0) {
      const float tdr = T[0].r, tdi = T[0].i;
      Fq[0].r = tdr + tdi;
      Fq[0].i = 0.f;
      Fq[NC].r = tdr - tdi;
      Fq[NC].i = 0.f;
    } else {
      const cpx fpk = T[k];
      cpx fpnk, f1k, f2k;
      fpnk.r = T[NC - k].r;
      fpnk.i = -T[NC - k].i;
      f1k.r = fpk.r + fpnk.r;
      f1k.i = fpk.i + fpnk.i;
      f2k.r = fpk.r - fpnk.r;
      f2k.i = fpk.i - fpnk.i;
      const cpx twv = cmul(f2k, sup_f[k - 1]);
      const float a_r = (f1k.r + twv.r) * 0.5f, a_i = (f1k.i + twv.i) * 0.5f;
      const float b_r = (f1k.r - twv.r) * 0.5f, b_i = (twv.i - f1k.i) * 0.5f;
      if (k != NC - k) {
        Fq[k].r = a_r;
        Fq[k].i = a_i;
      }
      Fq[NC - k].r = b_r;
      Fq[NC - k].i = b_i;
    }
  }
  __syncthreads();

  BTA_STAMP(3)
  if (threshold) {
    for (int w = tid; w < 128 * SQW; w += THREADS) {  // squared normalised real parts (.c:365-375)
      const int m = w % SQW, e = w / SQW;
      float v2 = 0.0f;
      if (m < NCOL) {
        const float v = coef[(e >> 4) * NB + 1 + m * 16 + (e & 15)].r * P.norm;
        v2 = v * v;
      }
      sq[w] = v2;
    }
    __syncthreads();
    BTA_STAMP(4)
    {  // SURE of the 15 segmentations of every macro-column (.c:354-401): bt_sure.h (lane = column, the two
       // half-waves share a segmentation's blocks), 32 columns per pass, the segmentations dealt to eight slots
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nwaves = THREADS / 64;
      for (int slot = wave; slot < 8; slot += nwaves)
        for (int c0 = 0; c0 < NCOL; c0 += 32)
          aspbt_sure::sure_slot<SQW>(slot, sq + c0, sure + c0 * 15, P, lane, min(NCOL - c0, 32));
    }
    __syncthreads();
    BTA_STAMP(5)
    // DC column and the bins past the last whole macro-column (.c:501-506, 518-532)
    for (int w = tid; w < 1 + (NB - (1 + NCOL * 16)); w += THREADS) {
      const int col = w == 0 ? 0 : (1 + NCOL * 16) + (w - 1);
      float sum = 0.0f;
      for (int t = 0; t < 8; ++t) {
        const float r = coef[t * NB + col].r, i = coef[t * NB + col].i;
        sum += r * r + i * i;
      }
      float a = 1 - P.dc_const / sum;
      if (a < 0) a = 0;
      for (int t = 0; t < 8; ++t) {
        thre[t * NB + col].r = coef[t * NB + col].r * a;
        thre[t * NB + col].i = coef[t * NB + col].i * a;
      }
    }
    {  // argmin (first wins, .c:404-416) + Stein attenuation of the chosen blocks (.c:421-454)
      const int wave = tid >> 6, lane = tid & 63, nwaves = THREADS / 64;
      float* pw = sure + NCOL * 15 + 16 + wave * (128 + 64);
      float* av = pw + 128;
      for (int m = wave; m < NCOL; m += nwaves) {
        const int base = 1 + m * 16;
        float best = sure[m * 15];
        int bc = 0;
        for (int c = 1; c < 15; ++c)
          if (sure[m * 15 + c] < best) {
            best = sure[m * 15 + c];
            bc = c;
          }
        const int T = bc / 5, F = bc % 5;
        cpx ce[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = lane + 64 * h;
          ce[h] = coef[(e >> 4) * NB + base + (e & 15)];
          pw[e] = ce[h].r * ce[h].r + ce[h].i * ce[h].i;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < (1 << (T + F))) {
          const float power = block_sum_dispatch<1>(bc, pw, lane >> F, lane & ((1 << F) - 1));
          float a = (float)(1.0 - (double)(P.seg[T][F].a_const / power));
          av[lane] = a * (float)(a > 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int e = lane + 64 * h, r = e >> 4, cc = e & 15;
          const float a = av[((r >> (3 - T)) << F) + (cc >> (4 - F))];
          thre[r * NB + base + cc].r = ce[h].r * a;
          thre[r * NB + base + cc].i = ce[h].i * a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
    BTA_STAMP(6)
    for (int w = tid; w < 8 * NC; w += THREADS) {  // empirical Wiener, Nyquist untouched (.c:469-486)
      const int t = fast_div(w, r_nc), f = w - t * NC;
      const float r = thre[t * NB + f].r, i = thre[t * NB + f].i;
      float wiener = r * r + i * i;
      wiener = wiener / (wiener + P.wiener_c);
      coef[t * NB + f].r *= wiener;
      coef[t * NB + f].i *= wiener;
    }
    __syncthreads();
  }

  BTA_STAMP(7)
  // ---- inverse STFT + overlap-add (blockThreshold_inverse_STFT, .c:284-300); kiss_fftri pre-pass
  // (kiss_fftr.c:137-157) straight into kiss order
  for (int i = tid; i < NC; i += THREADS) twl[i] = tw_i[i];  // the forward stages ended at a barrier long ago
  for (int w = tid; w < frames * (NC / 2 + 1); w += THREADS) {
    const int fr = fast_div(w, r_nh), k = w - fr * (NC / 2 + 1);
    const cpx* Fq = coef + fr * NB;
    cpx* T = work + fr * NC;
    if (k == 0) {
      cpx v;
      v.r = Fq[0].r + Fq[NC].r;
      v.i = Fq[0].r - Fq[NC].r;
      T[A.perm[0]] = v;
    } else {
      const cpx fk = Fq[k];
      cpx fnkc, fek, tmp;
      fnkc.r = Fq[NC - k].r;
      fnkc.i = -Fq[NC - k].i;
      fek.r = fk.r + fnkc.r;
      fek.i = fk.i + fnkc.i;
      tmp.r = fk.r - fnkc.r;
      tmp.i = fk.i - fnkc.i;
      const cpx fok = cmul(tmp, sup_i[k - 1]);
      cpx a, b;
      a.r = fek.r + fok.r;
      a.i = fek.i + fok.i;
      b.r = fek.r - fok.r;
      b.i = (fek.i - fok.i) * -1;
      if (k != NC - k) T[A.perm[k]] = a;
      T[A.perm[NC - k]] = b;
    }
  }
  __syncthreads();
  BTA_STAMP(8)
  any_stages<THREADS, GENERIC>(work, twl, A, frames, true, tid);
  BTA_STAMP(9)
  const float* td = reinterpret_cast<const float*>(work);  // frame fr sample j at fr * N + j
  const float fn = (float)N;
  for (int q = tid; q < total + HALF; q += THREADS) {
    const int t2 = fast_div(q, r_nc), t1 = t2 - 1;
    float v = q < HALF ? st[kAnyOffOutTail + q] : 0.0f;
    if (t1 >= 0 && t1 < frames) v += td[t1 * N + (q - HALF * t1)] / fn;
    if (t2 < frames) v += td[t2 * N + (q - HALF * t2)] / fn;
    if (q < total)
      y[q] = v;
    else
      stage[q - total] = v;
  }
  __syncthreads();
  for (int i = tid; i < HALF; i += THREADS) st[kAnyOffOutTail + i] = threshold ? stage[i] : 0.0f;
  BTA_STAMP(10)
#undef BTA_STAMP
}

// kiss_fftr / kiss_fftri seam for any even length: one workgroup per row
__global__ __launch_bounds__(kAnyThreads) void bt_fftr_any_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                   int inverse, const BtAnyTables* __restrict__ Ap) {
  const BtAnyTables& A = *Ap;
  const int NC = A.nc, N = A.n, tid = threadIdx.x;
  extern __shared__ __align__(16) unsigned char smem[];
  cpx* work = reinterpret_cast<cpx*>(smem);  // [NC]
  cpx* freq = work + NC;                     // [NC + 1]
  const cpx* tw = reinterpret_cast<const cpx*>(inverse ? A.tw_i : A.tw_f);
  const cpx* sup = reinterpret_cast<const cpx*>(inverse ? A.sup_i : A.sup_f);
  if (!inverse) {
    const float* x = src + (size_t)blockIdx.x * N;
    for (int n = tid; n < NC; n += kAnyThreads) work[A.perm[n]] = cpx{x[2 * n], x[2 * n + 1]};
    __syncthreads();
    any_stages<kAnyThreads, true>(work, tw, A, 1, false, tid);
    for (int k = tid; k < NC / 2 + 1; k += kAnyThreads) {
      if (k == 0) {
        const float tdr = work[0].r, tdi = work[0].i;
        freq[0] = cpx{tdr + tdi, 0.f};
        freq[NC] = cpx{tdr - tdi, 0.f};
      } else {
        const cpx fpk = work[k];
        cpx fpnk, f1k, f2k;
        fpnk.r = work[NC - k].r;
        fpnk.i = -work[NC - k].i;
        f1k.r = fpk.r + fpnk.r;
        f1k.i = fpk.i + fpnk.i;
        f2k.r = fpk.r - fpnk.r;
        f2k.i = fpk.i - fpnk.i;
        const cpx twv = cmul(f2k, sup[k - 1]);
        if (k != NC - k) freq[k] = cpx{(f1k.r + twv.r) * 0.5f, (f1k.i + twv.i) * 0.5f};
        freq[NC - k] = cpx{(f1k.r - twv.r) * 0.5f, (twv.i - f1k.i) * 0.5f};
      }
    }
    __syncthreads();
    float* y = dst + (size_t)blockIdx.x * 2 * (NC + 1);
    for (int i = tid; i < NC + 1; i += kAnyThreads) {
      y[2 * i] = freq[i].r;
      y[2 * i + 1] = freq[i].i;
    }
  } else {
    const float* f = src + (size_t)blockIdx.x * 2 * (NC + 1);
    for (int i = tid; i < NC + 1; i += kAnyThreads) freq[i] = cpx{f[2 * i], f[2 * i + 1]};
    __syncthreads();
    for (int k = tid; k < NC / 2 + 1; k += kAnyThreads) {
      if (k == 0) {
        work[A.perm[0]] = cpx{freq[0].r + freq[NC].r, freq[0].r - freq[NC].r};
      } else {
        const cpx fk = freq[k];
        cpx fnkc, fek, tmp;
        fnkc.r = freq[NC - k].r;
        fnkc.i = -freq[NC - k].i;
        fek.r = fk.r + fnkc.r;
        fek.i = fk.i + fnkc.i;
        tmp.r = fk.r - fnkc.r;
        tmp.i = fk.i - fnkc.i;
        const cpx fok = cmul(tmp, sup[k - 1]);
        if (k != NC - k) work[A.perm[k]] = cpx{fek.r + fok.r, fek.i + fok.i};
        work[A.perm[NC - k]] = cpx{fek.r - fok.r, (fek.i - fok.i) * -1};
      }
    }
    __syncthreads();
    any_stages<kAnyThreads, true>(work, tw, A, 1, true, tid);
    float* y = dst + (size_t)blockIdx.x * N;
    for (int n = tid; n < NC; n += kAnyThreads) {
      y[2 * n] = work[n].r;
      y[2 * n + 1] = work[n].i;
    }
  }
}

// kiss_fftr / kiss_fftri seam: one workgroup per row.
template <int N>
__global__ __launch_bounds__(N / 2) void bt_fftr_kernel(const float* __restrict__ src,
                                                        float* __restrict__ dst, int inverse,
                                                        const BtTables* __restrict__ Tb) {
  constexpr int NC = N / 2, NB = N / 2 + 1;
  __shared__ cpx work[NC];
  __shared__ cpx freq[NB];
  const int tid = threadIdx.x, row = blockIdx.x;
  const cpx* tw_f = reinterpret_cast<const cpx*>(N == 1024 ? Tb->tw1024_f : Tb->tw256_f);
  const cpx* tw_i = reinterpret_cast<const cpx*>(N == 1024 ? Tb->tw1024_i : Tb->tw256_i);
  const cpx* sup_f = reinterpret_cast<const cpx*>(N == 1024 ? Tb->sup1024_f : Tb->sup256_f);
  const cpx* sup_i = reinterpret_cast<const cpx*>(N == 1024 ? Tb->sup1024_i : Tb->sup256_i);
  if (!inverse) {
    const float* x = src + (size_t)row * N;
    cpx z;
    z.r = x[2 * tid];
    z.i = x[2 * tid + 1];
    work[kiss_position<NC>(tid)] = z;
    __syncthreads();
    kiss_stages<NC>(work, tw_f, 1, false, tid);
    real_split_forward<NC>(work, freq, sup_f, 1, tid);
    __syncthreads();
    float* o = dst + (size_t)row * 2 * NB;
    for (int k = tid; k < NB; k += NC) {
      o[2 * k] = freq[k].r;
      o[2 * k + 1] = freq[k].i;
    }
  } else {
    const float* f = src + (size_t)row * 2 * NB;
    for (int k = tid; k < NB; k += NC) {
      freq[k].r = f[2 * k];
      freq[k].i = f[2 * k + 1];
    }
    __syncthreads();
    real_merge_inverse<NC>(freq, work, sup_i, 1, tid);
    __syncthreads();
    kiss_stages<NC>(work, tw_i, 1, true, tid);
    float* o = dst + (size_t)row * N;
    o[2 * tid] = work[tid].r;
    o[2 * tid + 1] = work[tid].i;
  }
}

}  // namespace

namespace aspbt {

hipError_t launch_bt_macroblock8(float* state, const BtTables* T, const float* in, float* out,
                                 int num_streams, int in_stride, int out_stride, hipStream_t s,
                                 unsigned long long* stamps);
hipError_t launch_bt_macroblock8_q4(float* state, const BtTables* T, const float* in, float* out, int groups,
                                    int in_stride, int out_stride, hipStream_t s);
hipError_t launch_bt_fftr8(const float* src, float* dst, int count, int inverse, const BtTables* T,
                           hipStream_t s);

size_t macroblock_lds_bytes(int n) {
  const int nb = n / 2 + 1, ncol = (n - 1) / 2 / 16;
  const size_t need = (size_t)ncol * 15 + 16 + (size_t)(n / 2 / 64) * 192;
  size_t sure = (need > (size_t)n / 2 ? need : (size_t)n / 2) * sizeof(float);
  return (size_t)2 * 8 * nb * 8 + sure;
}

hipError_t launch_bt_macroblock(int n, float* state, const BtTables* T, const float* in,
                                float* out, int num_streams, int frames, int threshold,
                                int in_stride, int out_stride, hipStream_t s,
                                unsigned long long* stamps) {
  // whole macroblocks at N = 1024 take the one-wave-per-frame kernel (bt_kernels8.hip; 8-byte accesses)
  if (n == 1024 && frames == 8 && threshold == 1 && ((uintptr_t)in & 7) == 0 && ((uintptr_t)out & 7) == 0 &&
      (in_stride & 1) == 0 && (out_stride & 1) == 0 && !getenv("ASP_BT_OLD_KERNEL"))
    return launch_bt_macroblock8(state, T, in, out, num_streams, in_stride, out_stride, s, stamps);
  // whole macroblocks at N = 256: four stream-channels per workgroup in the same kernel; the last
  // num_streams % 4 stream-channels take the plain kernel below
  if (n == 256 && frames == 8 && threshold == 1 && num_streams >= 4 && ((uintptr_t)in & 7) == 0 &&
      ((uintptr_t)out & 7) == 0 && (in_stride & 1) == 0 && (out_stride & 1) == 0 && !getenv("ASP_BT_OLD_KERNEL")) {
    const int groups = num_streams / 4, rest = num_streams - 4 * groups;
    hipError_t e = launch_bt_macroblock8_q4(state, T, in, out, groups, in_stride, out_stride, s);
    if (e != hipSuccess || rest == 0) return e;
    const size_t done = (size_t)4 * groups;
    hipLaunchKernelGGL(bt_macroblock_kernel<256>, dim3(rest), dim3(128), macroblock_lds_bytes(256), s,
                       state + done * kStateFloats, T, in + done * in_stride, out + done * out_stride, frames, threshold,
                       in_stride, out_stride, (unsigned long long*)nullptr);
    return hipGetLastError();
  }
  const size_t lds = macroblock_lds_bytes(n);
  if (n == 1024) {
    static bool attr_set[64] = {};  // per device: the attribute belongs to the device's copy of the kernel
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bt_macroblock_kernel<1024>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr_set[dev] = true;
    }
    hipLaunchKernelGGL(bt_macroblock_kernel<1024>, dim3(num_streams), dim3(512), lds, s, state, T,
                       in, out, frames, threshold, in_stride, out_stride, stamps);
  } else {
    hipLaunchKernelGGL(bt_macroblock_kernel<256>, dim3(num_streams), dim3(128), lds, s, state, T,
                       in, out, frames, threshold, in_stride, out_stride, stamps);
  }
  return hipGetLastError();
}

hipError_t launch_bt_fftr(int n, const float* src, float* dst, int count, int inverse,
                          const BtTables* T, hipStream_t s) {
  if (n == 1024 && !getenv("ASP_BT_OLD_KERNEL")) return launch_bt_fftr8(src, dst, count, inverse, T, s);
  if (n == 1024)
    hipLaunchKernelGGL(bt_fftr_kernel<1024>, dim3(count), dim3(512), 0, s, src, dst, inverse, T);
  else
    hipLaunchKernelGGL(bt_fftr_kernel<256>, dim3(count), dim3(128), 0, s, src, dst, inverse, T);
  return hipGetLastError();
}

// any even window of 4 .. kAnyMaxWin samples (bt_macroblock_any_kernel); `state` has kAnyStateFloats per stream
hipError_t launch_bt_macroblock_any(const BtAnyTables& A, float* state, const float* in, float* out, int num_streams,
                                    int frames, int threshold, int in_stride, int out_stride, hipStream_t s,
                                    unsigned long long* stamps) {
  const int sqw = A.ncol <= 15 ? 16 : A.ncol <= 31 ? 32 : 64;  // narrow table: it fits inside the attenuated tile sooner
  const int threads = A.nc >= 192 ? 512 : 256;  // measured: 320-sample windows prefer 256 threads (five workgroups per CU), 480 and up 512
  const size_t tile = (size_t)8 * (A.nc + 1) * sizeof(cpx), sq = (size_t)128 * sqw * sizeof(float);
  const size_t lds = 2 * tile + (size_t)A.nc * sizeof(cpx) + (tile >= sq ? 0 : sq) +
                     (size_t)(A.ncol * 15 + 16 + (threads / 64) * (128 + 64) + A.nc) * sizeof(float);
  static bool attr_set[64] = {};  // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {  // the longest window needs 151 KB
    const int max_lds = 156 * 1024;
    hipError_t e = hipSuccess;
    const void* big[] = {reinterpret_cast<const void*>(bt_macroblock_any_kernel<512, 64, false>),
                         reinterpret_cast<const void*>(bt_macroblock_any_kernel<512, 64, true>),
                         reinterpret_cast<const void*>(bt_macroblock_any_kernel<512, 32, false>),
                         reinterpret_cast<const void*>(bt_macroblock_any_kernel<512, 32, true>)};
    for (const void* f : big)
      if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  bool generic = false;
  for (int q = 0; q < A.nfac; ++q) generic |= A.fac[2 * q] > 5;
#define BT_ANY_LAUNCH(TH, SQ)                                                                                  \
  do {                                                                                                         \
    if (generic)                                                                                               \
      hipLaunchKernelGGL((bt_macroblock_any_kernel<TH, SQ, true>), dim3(num_streams), dim3(TH), lds, s, state, A.self, in, \
                         out, frames, threshold, in_stride, out_stride, stamps);                               \
    else                                                                                                       \
      hipLaunchKernelGGL((bt_macroblock_any_kernel<TH, SQ, false>), dim3(num_streams), dim3(TH), lds, s, state, A.self, in, \
                         out, frames, threshold, in_stride, out_stride, stamps);                               \
  } while (0)
  if (threads == 512 && sqw == 64)
    BT_ANY_LAUNCH(512, 64);
  else if (threads == 512 && sqw == 32)
    BT_ANY_LAUNCH(512, 32);
  else if (threads == 512)
    BT_ANY_LAUNCH(512, 16);
  else if (sqw == 32)
    BT_ANY_LAUNCH(256, 32);
  else
    BT_ANY_LAUNCH(256, 16);
#undef BT_ANY_LAUNCH
  return hipGetLastError();
}

hipError_t launch_bt_fftr_any(const BtAnyTables& A, const float* src, float* dst, int count, int inverse, hipStream_t s) {
  hipLaunchKernelGGL(bt_fftr_any_kernel, dim3(count), dim3(kAnyThreads), (size_t)(2 * A.nc + 1) * sizeof(cpx), s, src, dst,
                     inverse, A.self);
  return hipGetLastError();
}

}  // namespace aspbt
