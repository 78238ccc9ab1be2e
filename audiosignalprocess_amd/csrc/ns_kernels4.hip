// ns_kernels4.hip -- the fused Analyze+Process frame step, one stream per wave64 (pair layout of
// ns_kernels1.hip), with the per-stream SCALAR work and the bin-128 chains of the four streams of a
// workgroup done ONCE, lane-parallel, by one of its four waves.
//
// Why: at 4096 streams the frame step is bound by vector-instruction issue (profiles/README.md,
// round 2: adding 480 VALU instructions per wave to ns_frame1_kernel costs 4.3 us per step).  In
// ns_frame1_kernel a third of a wave's instructions are wave-uniform: the feature / prior-model /
// gain-factor arithmetic of the stream (ns_core.c:523-790, 1315-1342: exp, tanh, sqrt, a dozen
// exact divisions, ten cross-bin sums) and the whole per-bin chain of bin 128, which every lane
// computes for one value.  Here the four streams of a workgroup put those values side by side:
//
//   * every wave does the per-bin work of its stream for the 128 bins its lanes own (two per
//     lane, no third slot), reduces each cross-bin sum to its four 16-lane row sums with four DPP
//     steps, and leaves the row sums in LDS;
//   * the workgroup's "scalar wave" (wave blockIdx & 3, so that the four workgroups of a CU load
//     different SIMDs) then runs the scalar section on lanes 0..3 = streams 0..3: finishes the sums
//     ((r0 + r1) + (r2 + r3), then + the bin-128 term), runs bin 128's quantile trackers, SNR,
//     likelihood ratio, noise update and gain, the feature updates, histograms, tanh maps, prior
//     model and gain factor -- the same float operations in the same order as ns_frame1_kernel, once
//     for four streams instead of four times for one -- and hands the three values the waves need
//     (the speech-probability prior gain, the gained R128, the energy gain factor) back through LDS;
//   * four workgroup barriers per frame (+ the table staging one) separate the phases.
//
// Cross-bin sums: lane-local (slot A + slot B), xor butterfly over the 64 lanes, bin 128 added LAST
// (ns_frame1_kernel adds it to lane 0's partial first); oracle/ns_oracle.c reproduces this as
// ASP_NS_REDUCE_TREE64Q and the tests compare bit for bit, outputs and every state array.
#include <hip/hip_runtime.h>

#include "ns_device.h"
#include "ns_layout.h"
#include "ns_pair_fft.h"

namespace {
using namespace aspns_dev;
using namespace aspns_pair;

// exchange slots of one stream (floats): wave -> scalar wave
enum : int {
  X_SE = 0, X_FL = 4, X_CV = 8, X_VP = 12, X_VM = 16, X_LL = 20, X_E2 = 24,  // 4 row sums each
  X_ENERGY1 = 28, X_R128, X_MAGN0, X_SUMMAGN, X_APSUM, X_LL127, X_SLM, X_SLILM,
  X_STRIDE = 40
};
// scalar wave -> wave
enum : int { O_GAINPRIOR = 0, O_RE128S, O_FACTOR, O_STRIDE = 4 };

// the four xor steps inside a 16-lane row: every lane of a row ends with the row's sum
__device__ __forceinline__ float row_sum(float v) {
  v = v + dpp_move<0xB1>(v);   // xor 1
  v = v + dpp_move<0x4E>(v);   // xor 2
  v = v + dpp_move<0x141>(v);  // xor 4
  v = v + dpp_move<0x140>(v);  // xor 8
  return v;
}
__device__ __forceinline__ float tree4(const float* r) {  // rows combined as the xor-16 / xor-32 steps do
  const float4 v = *reinterpret_cast<const float4*>(r);
  return (v.x + v.y) + (v.z + v.w);
}

#ifndef NS4_MIN_WAVES
#define NS4_MIN_WAVES 4
#endif

template <bool IO16>
__global__ __launch_bounds__(256, NS4_MIN_WAVES) void ns_frame4_kernel(float* __restrict__ state,
                                                           int32_t* __restrict__ hist_all,
                                                           const NsTables* __restrict__ T,
                                                           const float* __restrict__ in,
                                                           float* __restrict__ out,
                                                           int num_streams) {
  __shared__ float2 lds[4][128];
  __shared__ __align__(16) float tabs[3 * 64 * 4 + 32 * 4 * 2];
  __shared__ __align__(16) float wins[kAnal];
  __shared__ __align__(16) double exp2s[64];
  __shared__ __align__(16) double2 logts[128];
  __shared__ __align__(16) float xin[4][X_STRIDE];   // per stream: row sums and lane values for the scalar wave
  __shared__ __align__(16) float xout[4][O_STRIDE];  // per stream: what the scalar wave hands back
  __shared__ float scal[4][64];                      // per stream: the scalar row (slot k = scalar k)
  const int tid = threadIdx.x;
  // ---- prologue: every load of the first phase before the first wait (tables first)
  const float4* tab_src = tid < 192 ? reinterpret_cast<const float4*>(&T->tw[0][0][0]) + tid
                                    : reinterpret_cast<const float4*>(&T->spl[0][0][0]) + (tid - 192);
  const float4 tab_v = *tab_src;
  const float4 win_v = reinterpret_cast<const float4*>(T->window)[tid & 63];
  const double exp2_v = T->exp2_64[tid & 63];
  const double2 logt_v = reinterpret_cast<const double2*>(T->logtab)[tid & 127];
  const int lane = tid & 63;
  const int diagbits = T->diag[lane];
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int stream_raw = blockIdx.x * 4 + wv;
  const bool wave_live = stream_raw < num_streams;
  const int stream = wave_live ? stream_raw : num_streams - 1;  // clamped for the loads
  float* __restrict__ st = state + (size_t)stream * kStreamDwords;
  float* __restrict__ vec = st + kOffVec;
  float2* tile = lds[wv];
  const int lam = lane >> 1, h = lane & 1;
  const int g = lam >> 4, q = lam & 15;
  const int binA = q + 64 * g + 16 * h;  // slot 0; slot 1 = binA + 32
  const uint32_t gmask = g ? 0x80000000u : 0u;
  // which wave also runs the workgroup's scalar section: the one on SIMD (workgroup slot & 3) of the CU,
  // so that the (up to four) workgroups a CU holds put their scalar sections on different SIMDs;
  // elected through LDS (HW_ID: SIMD_ID bits 5:4, TG_ID bits 19:16), wave 0 if no wave sits there
  __shared__ int simd_of[4];
  const unsigned hw_simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_REG_HW_ID[5:4]
  const unsigned hw_tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4);    // HW_REG_HW_ID[19:16]
  if (lane == 0) simd_of[wv] = (int)hw_simd;

  const float sv = st[kOffScalars + lane];  // lane k holds scalar k of the wave's stream
#define SC_I(k) __builtin_amdgcn_readlane(__float_as_int(sv), (k))
#define SC_F(k) __int_as_float(SC_I(k))

  // ---- sliding analysis buffer [96 carried | 160 new]: lane L owns samples 4L .. 4L+3
  float* hbuf = st + kOffAnaHist;
  float4 s4;
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
#define LOADV(dst, f)                                                                          \
  {                                                                                            \
    const float2 v2_ = *reinterpret_cast<const float2*>(vec + (f)*kVecStride + 2 * lane);      \
    dst[0] = v2_.x; dst[1] = v2_.y;                                                            \
  }
#define STOREV(f, srcv) \
  *reinterpret_cast<float2*>(vec + (f)*kVecStride + 2 * lane) = make_float2(srcv[0], srcv[1]);
  const float2 carryA = *reinterpret_cast<const float2*>(st + kOffSynt + 2 * q + 32 * h);
  const float2 carryB = *reinterpret_cast<const float2*>(st + kOffSynt + 2 * q + 64);

  // ---- table staging (the loads above are in flight behind it)
  reinterpret_cast<float4*>(tabs)[tid] = tab_v;
  if (tid < 64) exp2s[tid] = exp2_v;
  if (tid < 128) logts[tid] = logt_v;
  if (tid >= 192) reinterpret_cast<float4*>(wins)[tid - 192] = win_v;
  scal[wv][lane] = sv;
  __syncthreads();
  int swave = 0;
  {
    const int want = (int)(hw_tg & 3u);
    if (simd_of[3] == want) swave = 3;
    if (simd_of[2] == want) swave = 2;
    if (simd_of[1] == want) swave = 1;
    if (simd_of[0] == want) swave = 0;
  }
  const bool is_s = wv == swave;
  const float* tws = tabs;
  const float* spls = tabs + 3 * 64 * 4;

  const float4 w4 = *reinterpret_cast<const float4*>(wins + 4 * lane);
  const float wx0 = w4.x * s4.x, wx1 = w4.y * s4.y, wx2 = w4.z * s4.z, wx3 = w4.w * s4.w;
  // Windowing + Energy (ns_core.c:969-978, 951-960)
  float epart = wx0 * wx0;
  epart += wx1 * wx1;
  epart += wx2 * wx2;
  epart += wx3 * wx3;
  const float energy1 = wave_sum(epart);
  const bool live = wave_live && energy1 != 0.0f;  // wave-uniform

  if (wave_live) {
    // the carried 96 samples of the next frame are this frame's last 96
    if (lane >= 40) *reinterpret_cast<float4*>(hbuf + 4 * (lane - 40)) = s4;
    if (energy1 == 0.0f) {
      // Analyze: nothing but the buffer slide (ns_core.c:1072-1082); Process: emit the synthesis
      // tail and clear it (ns_core.c:1239-1264).  The wave still meets the workgroup's barriers below.
      float* sy = st + kOffSynt;
      float* y = IO16 ? reinterpret_cast<float*>(reinterpret_cast<short*>(out) + (size_t)stream * kBlockL)
                      : out + (size_t)stream * kBlockL;
      float2 o01 = make_float2(0.f, 0.f);
      if (lane < 48) o01 = *reinterpret_cast<const float2*>(sy + 2 * lane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      store2p<IO16>(y, 2 * lane, sat16p(o01.x), sat16p(o01.y));
      if (lane < 16) store2p<IO16>(y, 128 + 2 * lane, 0.f, 0.f);
      if (lane < 48) *reinterpret_cast<float2*>(sy + 2 * lane) = make_float2(0.f, 0.f);
    }
  }

  // per-wave values that live across the barriers
  float LQ[3][2], DEN[3][2], quant[2];
  float smooth[2], noisePrev[2], magnPrevA[2], logLrt[2], avgPause[2];
  float re[2], im[2], magn[2], noise[2], prevStsa[2], snrLocPost[2], snrLocPrior[2];
  float R128 = 1.f;
  int blockInd = SC_I(S_BLOCKIND) + 1;  // ns_core.c:1084
  const float overdrive = SC_F(S_OVERDRIVE);
  const float denoiseBound = SC_F(S_DENOISEBOUND);
  const bool startup = blockInd < NS_END_STARTUP_SHORT;
  float2 el[2];

  if (live) {
    LOADV(LQ[0], V_LQ0) LOADV(LQ[1], V_LQ1) LOADV(LQ[2], V_LQ2)
    LOADV(DEN[0], V_DEN0) LOADV(DEN[1], V_DEN1) LOADV(DEN[2], V_DEN2)
    LOADV(quant, V_QUANT)
    // ---- forward FFT (ns_core.c:886-911)
    *reinterpret_cast<float4*>(&tile[2 * lane]) = make_float4(wx0, wx1, wx2, wx3);
    lds_sync1();
    cft128_passes1(tile, tws, diagbits, lane, el[0], el[1]);
    radix2_tail1(el[0], el[1], gmask, false);
    real_split1(tile, spls, lane, el, false);
    LOADV(magnPrevA, V_MAGNPREV_A) LOADV(logLrt, V_LOGLRT) LOADV(avgPause, V_AVGPAUSE)
    LOADV(smooth, V_SMOOTH) LOADV(noisePrev, V_NOISEPREV)

    re[0] = el[0].x;
    im[0] = el[0].y;
    re[1] = el[1].x;
    im[1] = el[1].y;
    R128 = lane_bcast(el[0].y, 0);  // R128 sits in the imaginary slot of element 0 (lane 0, slot 0)
    if (lane == 0) im[0] = 0.f;
    {
      float m2[2] = {re[0] * re[0] + im[0] * im[0], re[1] * re[1] + im[1] * im[1]}, rt[2];
      fsqrt_n<2>(m2, rt);
      magn[0] = rt[0] + 1.f;
      magn[1] = rt[1] + 1.f;
    }
    if (lane == 0) magn[0] = fabsf(re[0]) + 1.f;
    const float magnT = fabsf(R128) + 1.f;

    int updates = SC_I(S_UPDATES);
    int counter[3] = {SC_I(S_COUNTER0), SC_I(S_COUNTER1), SC_I(S_COUNTER2)};
    float lmagn[2];
    log_f32_via_tab_n<2>(magn, lmagn, logts);

    // sums the per-bin work below needs itself: finished in the wave (bin 128 added last)
    const float sumMagn = wave_sum(magn[0] + magn[1]) + magnT;
    const float apT = SC_F(S_TAIL0 + V_AVGPAUSE);
    const float apSum = wave_sum(avgPause[0] + avgPause[1]) + apT;
    const float avgMagn = DIV129(sumMagn), avgPauseMean = DIV129(apSum);
    // row sums for the scalar wave
    const float r_se = row_sum((re[0] * re[0] + im[0] * im[0]) + (re[1] * re[1] + im[1] * im[1]));
    const float r_fl = row_sum((lane == 0 ? 0.f : lmagn[0]) + lmagn[1]);  // bin 0 excluded (ns_core.c:538)
    float r_cv, r_vp, r_vm;
    {
      const float dm0 = magn[0] - avgMagn, dp0 = avgPause[0] - avgPauseMean;
      const float dm1 = magn[1] - avgMagn, dp1 = avgPause[1] - avgPauseMean;
      r_cv = row_sum(dm0 * dp0 + dm1 * dp1);
      r_vp = row_sum(dp0 * dp0 + dp1 * dp1);
      r_vm = row_sum(dm0 * dm0 + dm1 * dm1);
    }

    // ---- NoiseEstimation (ns_core.c:217-285), the lane's two bins
    if (updates < NS_END_STARTUP_LONG) updates++;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float cnt = (float)counter[s];
      const float cnt1 = (float)(counter[s] + 1);
      const float rcnt1 = fdiv(1.f, cnt1);
      {
        P2 den(DEN[s]), lq(LQ[s]);
        const P2 lm(lmagn);
        const P2 dq = fdiv2v(P2(NS_FACTOR * 1.f), den);  // used where density > 1
        const P2 delta = sel2(gt2(den, P2(1.0f)), dq, P2(NS_FACTOR));
        const B2 up = gt2(lm, lq);
        const P2 step = div_by_uniform2(sel2(up, NS_QUANTILE * delta, (1.f - NS_QUANTILE) * delta), cnt1, rcnt1);
        lq = sel2(up, lq + step, lq - step);
        const P2 nd = div_by_uniform2(cnt * den + 1.f / (2.f * NS_WIDTH), cnt1, rcnt1);
        den = sel2(lt2(abs2(lm - lq), P2(NS_WIDTH)), nd, den);
        den.store(DEN[s]);
        lq.store(LQ[s]);
      }
      if (counter[s] >= NS_END_STARTUP_LONG) {
        counter[s] = 0;
        if (updates >= NS_END_STARTUP_LONG) exp_f32_via_f64_n<2>(LQ[s], quant, exp2s);
      }
      counter[s]++;
    }
    if (updates < NS_END_STARTUP_LONG) exp_f32_via_f64_n<2>(LQ[2], quant, exp2s);
    noise[0] = quant[0];
    noise[1] = quant[1];
    STOREV(V_LQ0, LQ[0]) STOREV(V_LQ1, LQ[1]) STOREV(V_LQ2, LQ[2])
    STOREV(V_DEN0, DEN[0]) STOREV(V_DEN1, DEN[1]) STOREV(V_DEN2, DEN[2])
    STOREV(V_QUANT, quant)

    // ---- startup noise model (ns_core.c:1091-1100, 1109-1162): the first 50 blocks of a stream.
    // The model's scalars are recomputed here from the same inputs the scalar wave uses (rare path).
    float slm = 0.f, slilm = 0.f;
    if (startup) {
      const float lmagnT = log_f32_via_tab(magnT, logts);
      float lm2[2], lilm[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int bin = binA + 32 * k;
        const float li = T->logi[bin];
        lm2[k] = bin >= NS_START_BAND ? lmagn[k] : 0.f;
        lilm[k] = bin >= NS_START_BAND ? li * lmagn[k] : 0.f;
      }
      slm = wave_sum(lm2[0] + lm2[1]) + lmagnT;
      slilm = wave_sum(lilm[0] + lilm[1]) + T->logi[128] * lmagnT;
      const float sum_log_i = T->sum_log_i, sum_log_i_square = T->sum_log_i_square;
      float whiteNoiseLevel = SC_F(S_WHITE), pinkNoiseNumerator = SC_F(S_PINKNUM), pinkNoiseExp = SC_F(S_PINKEXP);
      whiteNoiseLevel += DIV129(sumMagn) * overdrive;
      float tmpFloat1 = sum_log_i_square * ((float)(kBins - NS_START_BAND));
      tmpFloat1 -= (sum_log_i * sum_log_i);
      float tmpFloat2 = (sum_log_i_square * slm - sum_log_i * slilm);
      float tmpFloat3 = tmpFloat2 / tmpFloat1;
      if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
      pinkNoiseNumerator += tmpFloat3;
      tmpFloat2 = (sum_log_i * slm);
      tmpFloat2 -= ((float)(kBins - NS_START_BAND)) * slilm;
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
      float pn[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int bin = binA + 32 * k;
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
      STOREV(V_PARAMNOISE, pn)
    }

    // ---- ComputeSnr (ns_core.c:566-588)
    {
      float dn1[2], dn2[2], q1[2], q2[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        dn1[k] = noisePrev[k] + 0.0001f;
        dn2[k] = noise[k] + 0.0001f;
      }
      fdiv2a(magnPrevA, dn1, q1);
      fdiv2a(magn, dn2, q2);  // used where magn > noise
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float previousEstimateStsa = q1[k] * smooth[k];
        prevStsa[k] = previousEstimateStsa;
        snrLocPost[k] = 0.f;
        if (magn[k] > noise[k]) snrLocPost[k] = q2[k] - 1.f;
        snrLocPrior[k] = NS_DD_PR_SNR * previousEstimateStsa + (1.f - NS_DD_PR_SNR) * snrLocPost[k];
      }
    }
    // ---- SpeechNoiseProb, the per-bin likelihood-ratio update (ns_core.c:667-684)
    {
      float t1[2], lt1[2], tn[2], td2[2], t2v[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) t1[k] = 1.f + 2.f * snrLocPrior[k];
      log_f32_via_tab_n<2>(t1, lt1, logts);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        tn[k] = 2.f * snrLocPrior[k];
        td2[k] = t1[k] + 0.0001f;
      }
      fdiv2a(tn, td2, t2v);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float besselTmp = (snrLocPost[k] + 1.f) * t2v[k];
        logLrt[k] += NS_LRT_TAVG * (besselTmp - lt1[k] - logLrt[k]);
      }
    }
    const float r_ll = row_sum(logLrt[0] + logLrt[1]);

    // ---- hand the row sums and lane values to the scalar wave
    float* xi = xin[wv];
    if ((lane & 15) == 0) {
      const int r = lane >> 4;
      xi[X_SE + r] = r_se;
      xi[X_FL + r] = r_fl;
      xi[X_CV + r] = r_cv;
      xi[X_VP + r] = r_vp;
      xi[X_VM + r] = r_vm;
      xi[X_LL + r] = r_ll;
    }
    if (lane == 0) {
      xi[X_ENERGY1] = energy1;
      xi[X_R128] = R128;
      xi[X_MAGN0] = magn[0];
      xi[X_SUMMAGN] = sumMagn;
      xi[X_APSUM] = apSum;
      xi[X_SLM] = slm;
      xi[X_SLILM] = slilm;
    }
    if (lane == 63) xi[X_LL127] = logLrt[1];  // bin 127 = q 15, g 1, t 3: slot 1 of lane 63
  } else {
    // a silent (or absent) stream: benign values, the scalar wave stores nothing for it
    float* xi = xin[wv];
    if (lane < X_STRIDE) xi[lane] = lane == X_ENERGY1 ? 0.f : 1.f;
  }
  __syncthreads();  // ---------------------------------------------------------------- barrier 1

  // ==== the scalar section: lanes 0..3 (and their copies lane & 3) = streams 0..3 of the workgroup
  // values the scalar wave keeps from its first part to its second (after barrier 3)
  float s_priorSpeechProb = 0.f, s_energy1 = 0.f, s_denoiseBound = 0.f;
  int s_gainmap = 0, s_blockInd = 0;
  bool s_live = false;
  if (is_s) {
    const int ss = lane & 3;
    const int sstream_raw = blockIdx.x * 4 + ss;
    const int sstream = sstream_raw < num_streams ? sstream_raw : num_streams - 1;
    int32_t* __restrict__ shist = hist_all + (size_t)sstream * kHistDwords;
    const float* xi = xin[ss];
    float* sc = scal[ss];
#define SL_F(k) sc[(k)]
#define SL_I(k) __float_as_int(sc[(k)])
    const float energy1s = xi[X_ENERGY1];
    const bool slive = sstream_raw < num_streams && energy1s != 0.0f;
    const bool swrite = slive && lane < 4;
    int sblockInd = SL_I(S_BLOCKIND) + 1;
    const float soverdrive = SL_F(S_OVERDRIVE);
    const float sdenoiseBound = SL_F(S_DENOISEBOUND);
    float priorSpeechProb = SL_F(S_PRIORSPEECHPROB);
    const int updateParsFlag = SL_I(S_MUP0);
    int updates = SL_I(S_UPDATES);
    int counter[3] = {SL_I(S_COUNTER0), SL_I(S_COUNTER1), SL_I(S_COUNTER2)};
    const bool sstartup = sblockInd < NS_END_STARTUP_SHORT;
    // bin 128 of the state rows
    float LQt[3] = {SL_F(S_TAIL0 + V_LQ0), SL_F(S_TAIL0 + V_LQ1), SL_F(S_TAIL0 + V_LQ2)};
    float DENt[3] = {SL_F(S_TAIL0 + V_DEN0), SL_F(S_TAIL0 + V_DEN1), SL_F(S_TAIL0 + V_DEN2)};
    float quantT = SL_F(S_TAIL0 + V_QUANT);
    const float smoothT = SL_F(S_TAIL0 + V_SMOOTH), noisePrevT = SL_F(S_TAIL0 + V_NOISEPREV);
    const float magnPrevT = SL_F(S_TAIL0 + V_MAGNPREV_A);
    float logLrtT = SL_F(S_TAIL0 + V_LOGLRT), avgPauseT = SL_F(S_TAIL0 + V_AVGPAUSE);

    const float sR128 = xi[X_R128];
    const float magnT = fabsf(sR128) + 1.f;
    const float lmagnT = log_f32_via_tab(magnT, logts);
    float signalEnergy = tree4(xi + X_SE) + sR128 * sR128;
    signalEnergy = DIV129(signalEnergy);
    const float sumMagn = xi[X_SUMMAGN];

    // ---- NoiseEstimation for bin 128 (ns_core.c:217-285)
    if (updates < NS_END_STARTUP_LONG) updates++;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float cnt = (float)counter[s];
      const float cnt1 = (float)(counter[s] + 1);
      const float rcnt1 = fdiv(1.f, cnt1);
      {
        float den = DENt[s], lq = LQt[s];
        const float dq = fdiv(NS_FACTOR * 1.f, den);
        const float delta = den > 1.0f ? dq : NS_FACTOR;
        const bool up = lmagnT > lq;
        const float step = div_by_uniform(up ? NS_QUANTILE * delta : (1.f - NS_QUANTILE) * delta, cnt1, rcnt1);
        lq = up ? lq + step : lq - step;
        const float nd = div_by_uniform(cnt * den + 1.f / (2.f * NS_WIDTH), cnt1, rcnt1);
        den = fabsf(lmagnT - lq) < NS_WIDTH ? nd : den;
        DENt[s] = den;
        LQt[s] = lq;
      }
      if (counter[s] >= NS_END_STARTUP_LONG) {
        counter[s] = 0;
        if (updates >= NS_END_STARTUP_LONG) quantT = exp_f32_via_f64(LQt[s], exp2s);
      }
      counter[s]++;
    }
    if (updates < NS_END_STARTUP_LONG) quantT = exp_f32_via_f64(LQt[2], exp2s);
    float noiseT = quantT;

    // ---- startup noise model (ns_core.c:1091-1100, 1109-1162)
    float whiteNoiseLevel = SL_F(S_WHITE);
    float pinkNoiseNumerator = SL_F(S_PINKNUM);
    float pinkNoiseExp = SL_F(S_PINKEXP);
    float fd5 = SL_F(S_FD5);
    float pnT = SL_F(S_TAIL0 + V_PARAMNOISE);
    if (sstartup) {
      const float slm = xi[X_SLM], slilm = xi[X_SLILM];
      const float sum_log_i = T->sum_log_i, sum_log_i_square = T->sum_log_i_square;
      whiteNoiseLevel += DIV129(sumMagn) * soverdrive;
      float tmpFloat1 = sum_log_i_square * ((float)(kBins - NS_START_BAND));
      tmpFloat1 -= (sum_log_i * sum_log_i);
      float tmpFloat2 = (sum_log_i_square * slm - sum_log_i * slilm);
      float tmpFloat3 = tmpFloat2 / tmpFloat1;
      if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
      pinkNoiseNumerator += tmpFloat3;
      tmpFloat2 = (sum_log_i * slm);
      tmpFloat2 -= ((float)(kBins - NS_START_BAND)) * slilm;
      tmpFloat3 = tmpFloat2 / tmpFloat1;
      if (tmpFloat3 < 0.f) tmpFloat3 = 0.f;
      if (tmpFloat3 > 1.f) tmpFloat3 = 1.f;
      pinkNoiseExp += tmpFloat3;
      float parametric_num = 0.f, parametric_exp = 0.f;
      if (pinkNoiseExp > 0.f) {
        parametric_num = (float)exp((double)(pinkNoiseNumerator / (float)(sblockInd + 1)));
        parametric_num *= (float)(sblockInd + 1);
        parametric_exp = pinkNoiseExp / (float)(sblockInd + 1);
      }
      if (pinkNoiseExp == 0.f) {
        pnT = whiteNoiseLevel;
      } else {
        pnT = (float)((double)parametric_num / pow((double)128.0f, (double)parametric_exp));
      }
      noiseT *= (sblockInd);
      const float t2 = pnT * (NS_END_STARTUP_SHORT - sblockInd);
      noiseT += (t2 / (float)(sblockInd + 1));
      noiseT /= NS_END_STARTUP_SHORT;
    }
    if (sblockInd < NS_END_STARTUP_LONG) {  // ns_core.c:1165-1169
      fd5 *= sblockInd;
      fd5 += signalEnergy;
      fd5 /= (sblockInd + 1);
    }

    // ---- ComputeSnr for bin 128 (ns_core.c:566-588)
    const float prevStsaT = fdiv(magnPrevT, noisePrevT + 0.0001f) * smoothT;
    float snrPostT = 0.f;
    {
      const float q2 = fdiv(magnT, noiseT + 0.0001f);
      if (magnT > noiseT) snrPostT = q2 - 1.f;
    }
    const float snrPriorT = NS_DD_PR_SNR * prevStsaT + (1.f - NS_DD_PR_SNR) * snrPostT;

    // ---- ComputeSpectralFlatness (ns_core.c:523-556)
    float fd0 = SL_F(S_FD0), fd4 = SL_F(S_FD4), fd6 = SL_F(S_FD6);
    {
      float num = tree4(xi + X_FL) + lmagnT;
      float den = sumMagn - xi[X_MAGN0];
      den = DIV129(den);
      num = DIV129(num);
      const float spectralTmp = fdiv(exp_f32_via_f64(num, exp2s), den);
      fd0 += NS_SPECT_FL_TAVG * (spectralTmp - fd0);
    }
    // ---- ComputeSpectralDifference (ns_core.c:595-634)
    {
      float avgPauseMean = xi[X_APSUM];
      float avgMagn = sumMagn;
      avgPauseMean = DIV129(avgPauseMean);
      avgMagn = DIV129(avgMagn);
      const float dm = magnT - avgMagn, dp = avgPauseT - avgPauseMean;
      float covMagnPause = tree4(xi + X_CV) + dm * dp;
      float varPause = tree4(xi + X_VP) + dp * dp;
      float varMagn = tree4(xi + X_VM) + dm * dm;
      covMagnPause = DIV129(covMagnPause);
      varPause = DIV129(varPause);
      varMagn = DIV129(varMagn);
      fd6 += signalEnergy;
      float avgDiffNormMagn = varMagn - fdiv(covMagnPause * covMagnPause, varPause + 0.0001f);
      avgDiffNormMagn = fdiv(avgDiffNormMagn, fd5 + 0.0001f);
      fd4 += NS_SPECT_DIFF_TAVG * (avgDiffNormMagn - fd4);
    }

    // ---- histograms / prior model (FeatureUpdate, ns_core.c:766-790)
    float fd3 = SL_F(S_FD3);  // previous frame's average LRT feeds the histogram
    PriorModel pm;
    pm.p0 = SL_F(S_PMP0);
    pm.p1 = SL_F(S_PMP1);
    pm.p3 = SL_F(S_PMP3);
    pm.p4 = SL_F(S_PMP4);
    pm.p5 = SL_F(S_PMP5);
    pm.p6 = SL_F(S_PMP6);
    const float pmp2 = SL_F(S_PMP2);
    int mup0 = updateParsFlag, mup3 = SL_I(S_MUP3);
    const int mup1 = SL_I(S_MUP1);
    bool window_closed = false;
    if (updateParsFlag >= 1) {
      mup3--;
      if (mup3 > 0 && swrite) {
        // FeatureParameterExtraction(self, 0), ns_core.c:309-334: one writer per bin and stream
        if ((fd3 < kHist * 0.1f) && (fd3 >= 0.0f))
          atomicAdd(&shist[(int)div_by_uniform(fd3, 0.1f, 1.0f / 0.1f)], 1);
        if ((fd0 < kHist * 0.05f) && (fd0 >= 0.0f))
          atomicAdd(&shist[kHistStride + (int)div_by_uniform(fd0, 0.05f, 1.0f / 0.05f)], 1);
        if ((fd4 < kHist * 0.1f) && (fd4 >= 0.0f))
          atomicAdd(&shist[2 * kHistStride + (int)div_by_uniform(fd4, 0.1f, 1.0f / 0.1f)], 1);
      }
    }
    {
      // the window close needs the whole wave for one stream: stream by stream (once in 500 frames)
      const bool closing = updateParsFlag >= 1 && mup3 == 0 && slive;
      const unsigned long long closing_mask = __ballot(closing);
#pragma unroll
      for (int hs = 0; hs < 4; ++hs) {
        if (((closing_mask >> hs) & 1ull) != 0) {  // wave-uniform
          PriorModel pin;
          pin.p0 = __shfl(pm.p0, hs, 64);
          pin.p1 = __shfl(pm.p1, hs, 64);
          pin.p3 = __shfl(pm.p3, hs, 64);
          pin.p4 = __shfl(pm.p4, hs, 64);
          pin.p5 = __shfl(pm.p5, hs, 64);
          pin.p6 = __shfl(pm.p6, hs, 64);
          const int w1 = __shfl(mup1, hs, 64), f0 = __shfl(mup0, hs, 64);
          int32_t* hh = hist_all + (size_t)(blockIdx.x * 4 + hs) * kHistDwords;
          const PriorModel po = close_histogram_window(hh, lane, w1, f0 >= 1, pin);
          if (ss == hs) pm = po;
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

    // ---- SpeechNoiseProb (ns_core.c:642-749): bin 128's likelihood ratio, then the stream's prior
    {
      const float t1 = 1.f + 2.f * snrPriorT;
      const float lt1 = log_f32_via_tab(t1, logts);
      const float t2 = fdiv(2.f * snrPriorT, t1 + 0.0001f);
      const float besselTmp = (snrPostT + 1.f) * t2;
      logLrtT += NS_LRT_TAVG * (besselTmp - lt1 - logLrtT);
    }
    float logLrtTimeAvgKsum = tree4(xi + X_LL) + logLrtT;
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
      // the three tanh() of :696-725 for the four streams in one evaluation: lane 4 j + s carries
      // argument j of stream s (lanes >= 12 repeat argument 2)
      const int jj = lane >> 2;
      const float arg = jj == 0 ? arg0 : (jj == 1 ? arg1 : arg2);
      const float th = tanh_f32_via_f64(arg, exp2s);
      const float indicator0 = 0.5f * (__shfl(th, ss, 64) + 1.f);
      const float indicator1 = 0.5f * (__shfl(th, 4 + ss, 64) + 1.f);
      const float indicator2 = 0.5f * (__shfl(th, 8 + ss, 64) + 1.f);
      const float indPrior = pm.p4 * indicator0 + pm.p5 * indicator1 + pm.p6 * indicator2;
      priorSpeechProb += NS_PRIOR_UPDATE * (indPrior - priorSpeechProb);
      if (priorSpeechProb > 1.f) priorSpeechProb = 1.f;
      if (priorSpeechProb < 0.01f) priorSpeechProb = 0.01f;
    }
    const float gainPrior = fdiv(1.f - priorSpeechProb, priorSpeechProb + 0.0001f);
    // speech probability of bin 128 and of its predecessor bin 127 (ns_core.c:743-748)
    float probT, prob127;
    {
      float nl[2] = {-logLrtT, -xi[X_LL127]}, ev[2];
      exp_f32_via_f64_n<2>(nl, ev, exp2s);
      float pd[2];
      const float ones[2] = {1.f, 1.f};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float invLrt = ev[k];
        invLrt = (float)gainPrior * invLrt;
        pd[k] = 1.f + invLrt;
      }
      float pr[2];
      fdiv2a(ones, pd, pr);
      probT = pr[0];
      prob127 = pr[1];
    }
    // ---- UpdateNoiseEstimate for bin 128 (ns_core.c:800-846)
    {
      const float gammaOld = prob127 > NS_PROB_RANGE ? NS_SPEECH_UPDATE : NS_NOISE_UPDATE;
      const float ps = probT, pns = 1.f - probT;
      const float noiseUpdateTmp =
          gammaOld * noisePrevT + (1.f - gammaOld) * (pns * magnT + ps * noisePrevT);
      float gammaNew = NS_NOISE_UPDATE;
      if (ps > NS_PROB_RANGE) gammaNew = NS_SPEECH_UPDATE;
      if (ps < NS_PROB_RANGE) avgPauseT += NS_GAMMA_PAUSE * (magnT - avgPauseT);
      float nz;
      if (gammaNew == gammaOld) {
        nz = noiseUpdateTmp;
      } else {
        nz = gammaNew * noisePrevT + (1.f - gammaNew) * (pns * magnT + ps * noisePrevT);
        if (noiseUpdateTmp < nz) nz = noiseUpdateTmp;
      }
      noiseT = nz;
    }
    // ---- decision-directed Wiener gain of bin 128 (ns_core.c:985-1007, 1276-1307)
    float initMagnT = SL_F(S_TAIL0 + V_INITMAGN);
    if (sstartup) initMagnT += magnT;  // ns_core.c:1268-1272
    float gainT;
    {
      const float gq1 = fdiv(magnT, noiseT + 0.0001f);
      float currentEstimateStsa = 0.f;
      if (magnT > noiseT) currentEstimateStsa = gq1 - 1.f;
      const float snrP = NS_DD_PR_SNR * prevStsaT + (1.f - NS_DD_PR_SNR) * currentEstimateStsa;
      float gg = fdiv(snrP, soverdrive + snrP);
      if (gg < sdenoiseBound) gg = sdenoiseBound;
      if (gg > 1.f) gg = 1.f;
      if (sstartup) {
        float tmp = (initMagnT - soverdrive * pnT);
        tmp /= (initMagnT + 0.0001f);
        if (tmp < sdenoiseBound) tmp = sdenoiseBound;
        if (tmp > 1.f) tmp = 1.f;
        gg *= (sblockInd);
        tmp *= (NS_END_STARTUP_SHORT - sblockInd);
        gg += tmp;
        gg /= (NS_END_STARTUP_SHORT);
      }
      gainT = gg;
    }
    if (lane < 4) {
      xout[ss][O_GAINPRIOR] = gainPrior;
      xout[ss][O_RE128S] = sR128 * gainT;
    }
    // ---- commit the stream's scalars and bin-128 values to its row in LDS (live streams only)
    if (swrite) {
      sc[S_UPDATES] = __int_as_float(updates);
      sc[S_COUNTER0] = __int_as_float(counter[0]);
      sc[S_COUNTER1] = __int_as_float(counter[1]);
      sc[S_COUNTER2] = __int_as_float(counter[2]);
      sc[S_MUP0] = __int_as_float(mup0);
      sc[S_MUP3] = __int_as_float(mup3);
      sc[S_SIGNALENERGY] = signalEnergy;
      sc[S_SUMMAGN] = sumMagn;
      if (sstartup) {
        sc[S_WHITE] = whiteNoiseLevel;
        sc[S_PINKNUM] = pinkNoiseNumerator;
        sc[S_PINKEXP] = pinkNoiseExp;
        sc[S_TAIL0 + V_PARAMNOISE] = pnT;
        sc[S_TAIL0 + V_INITMAGN] = initMagnT;
      }
      if (window_closed) {
        sc[S_PMP0] = pm.p0;
        sc[S_PMP1] = pm.p1;
        sc[S_PMP3] = pm.p3;
        sc[S_PMP4] = pm.p4;
        sc[S_PMP5] = pm.p5;
        sc[S_PMP6] = pm.p6;
      }
      sc[S_FD0] = fd0;
      sc[S_FD3] = fd3;
      sc[S_FD4] = fd4;
      sc[S_FD5] = fd5;
      sc[S_FD6] = fd6;
      sc[S_BLOCKIND] = __int_as_float(sblockInd);
      sc[S_PRIORSPEECHPROB] = priorSpeechProb;
      sc[S_TAIL0 + V_LQ0] = LQt[0];
      sc[S_TAIL0 + V_LQ1] = LQt[1];
      sc[S_TAIL0 + V_LQ2] = LQt[2];
      sc[S_TAIL0 + V_DEN0] = DENt[0];
      sc[S_TAIL0 + V_DEN1] = DENt[1];
      sc[S_TAIL0 + V_DEN2] = DENt[2];
      sc[S_TAIL0 + V_QUANT] = quantT;
      sc[S_TAIL0 + V_LOGLRT] = logLrtT;
      sc[S_TAIL0 + V_AVGPAUSE] = avgPauseT;
      sc[S_TAIL0 + V_MAGNPREV_A] = magnT;     // ns_core.c:1180
      sc[S_TAIL0 + V_SMOOTH] = gainT;         // ns_core.c:1304
      sc[S_TAIL0 + V_NOISEPREV] = noiseT;     // ns_core.c:1310
    }
    s_priorSpeechProb = priorSpeechProb;
    s_energy1 = energy1s;
    s_denoiseBound = sdenoiseBound;
    s_gainmap = SL_I(S_GAINMAP);
    s_blockInd = sblockInd;
    s_live = slive;
#undef SL_F
#undef SL_I
  }
  __syncthreads();  // ---------------------------------------------------------------- barrier 2

  float td0 = 0.f, td1 = 0.f, td2 = 0.f, td3s = 0.f;
  if (live) {
    const float gainPrior = xout[wv][O_GAINPRIOR];
    const float re128s = xout[wv][O_RE128S];
    float probSpeech[2];
    {
      float nl[2] = {-logLrt[0], -logLrt[1]}, ev[2], pd[2];
      exp_f32_via_f64_n<2>(nl, ev, exp2s);
      const float ones[2] = {1.f, 1.f};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float invLrt = ev[k];
        invLrt = (float)gainPrior * invLrt;
        pd[k] = 1.f + invLrt;
      }
      fdiv2a(ones, pd, probSpeech);
    }
    // ---- UpdateNoiseEstimate (ns_core.c:800-846): the time constant carried into bin i is the one
    // bin i-1 selected (source lanes / slots as in ns_kernels1.hip)
    {
      const int srcA = q > 0 ? lane - 2 : (h ? 30 + 32 * g : 31);
      const int srcB = q > 0 ? lane - 2 : (h ? 30 + 32 * g : 31 + 32 * g);
      const bool a_from1 = q == 0 && h == 0;
      const bool b_from1 = q > 0 || h == 1;
      const float a0 = __shfl(probSpeech[0], srcA, 64), a1 = __shfl(probSpeech[1], srcA, 64);
      const float b0 = __shfl(probSpeech[0], srcB, 64), b1 = __shfl(probSpeech[1], srcB, 64);
      float prevProb[2];
      prevProb[0] = a_from1 ? a1 : a0;
      prevProb[1] = b_from1 ? b1 : b0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float gammaOld = prevProb[k] > NS_PROB_RANGE ? NS_SPEECH_UPDATE : NS_NOISE_UPDATE;
        if (k == 0 && lane == 0) gammaOld = NS_NOISE_UPDATE;  // bin 0 has no predecessor
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
    STOREV(V_LOGLRT, logLrt) STOREV(V_AVGPAUSE, avgPause)
    STOREV(V_MAGNPREV_A, magn)  // ns_core.c:1180 (== magnPrevProcess while paired)

    // ---- Process: decision-directed Wiener gain (ns_core.c:985-1007, 1276-1307)
    float initMagn[2], pnoise[2];
    if (startup) {  // ns_core.c:1268-1272
      LOADV(initMagn, V_INITMAGN)
      LOADV(pnoise, V_PARAMNOISE)
      initMagn[0] += magn[0];
      initMagn[1] += magn[1];
      STOREV(V_INITMAGN, initMagn)
    }
    float gainv[2], gq1[2], gq2[2], snrP[2];
    {
      float gd1[2], gd2[2];
      gd1[0] = noise[0] + 0.0001f;
      gd1[1] = noise[1] + 0.0001f;
      fdiv2a(magn, gd1, gq1);  // used where magn > noise
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float currentEstimateStsa = 0.f;
        if (magn[k] > noise[k]) currentEstimateStsa = gq1[k] - 1.f;
        snrP[k] = NS_DD_PR_SNR * prevStsa[k] + (1.f - NS_DD_PR_SNR) * currentEstimateStsa;
        gd2[k] = overdrive + snrP[k];
      }
      fdiv2a(snrP, gd2, gq2);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
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
    STOREV(V_SMOOTH, gainv)      // ns_core.c:1304
    STOREV(V_NOISEPREV, noise)   // ns_core.c:1310

    // ---- IFFT (ns_core.c:923-944)
    el[0] = make_float2(re[0], im[0]);
    el[1] = make_float2(re[1], im[1]);
    if (lane == 0) el[0].y = re128s;  // Ooura packing: a[1] = R128 (gained)
    real_split1(tile, spls, lane, el, true);
    lds_sync1();
    {
      const int base = 64 * g + q + 16 * h;
      tile[base] = el[0];
      tile[base + 32] = el[1];
    }
    lds_sync1();
    cft128_passes1(tile, tws, diagbits, lane, el[0], el[1]);
    radix2_tail1(el[0], el[1], gmask, true);
    td0 = el[0].x * (2.f / kAnal);
    td1 = el[0].y * (2.f / kAnal);
    td2 = el[1].x * (2.f / kAnal);
    td3s = el[1].y * (2.f / kAnal);
    // energy after the suppression (ns_core.c:1318-1321): row sums for the scalar wave
    float e2 = td0 * td0;
    e2 += td1 * td1;
    e2 += td2 * td2;
    e2 += td3s * td3s;
    const float r_e2 = row_sum(e2);
    if ((lane & 15) == 0) xin[wv][X_E2 + (lane >> 4)] = r_e2;
  }
  __syncthreads();  // ---------------------------------------------------------------- barrier 3

  if (is_s) {
    const int ss = lane & 3;
    // ---- energy-based gain compensation (ns_core.c:1315-1342)
    float factor = 1.f;
    if (s_gainmap == 1 && s_blockInd > NS_END_STARTUP_LONG) {
      float factor1 = 1.f, factor2 = 1.f;
      const float energy2 = tree4(xin[ss] + X_E2);
      float gain = fsqrt(fdiv(energy2, s_energy1 + 1.f));
      if (gain > NS_B_LIM) {
        factor1 = 1.f + 1.3f * (gain - NS_B_LIM);
        if (gain * factor1 > 1.f) factor1 = fdiv(1.f, gain);
      }
      if (gain < NS_B_LIM) {
        if (gain <= s_denoiseBound) gain = s_denoiseBound;
        factor2 = 1.f - 0.3f * (NS_B_LIM - gain);
      }
      factor = s_priorSpeechProb * factor1 + (1.f - s_priorSpeechProb) * factor2;
    }
    if (lane < 4) xout[ss][O_FACTOR] = factor;
    (void)s_live;
  }
  __syncthreads();  // ---------------------------------------------------------------- barrier 4

  if (live) {
    const float factor = xout[wv][O_FACTOR];
    // ---- synthesis window, overlap-add, emit 160, carry 96 (ns_core.c:1344-1359)
    float* sy = st + kOffSynt;
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
      *reinterpret_cast<float2*>(sy + nA - 160) = make_float2(oA0, oA1);
    } else {
      store2p<IO16>(y, nA, sat16p(oA0), sat16p(oA1));
    }
    if (nB >= 160) {
      *reinterpret_cast<float2*>(sy + nB - 160) = make_float2(oB0, oB1);
    } else {
      store2p<IO16>(y, nB, sat16p(oB0), sat16p(oB1));
    }
    // the stream's scalar row, as the scalar wave left it
    st[kOffScalars + lane] = scal[wv][lane];
  }
#undef SC_I
#undef SC_F
#undef LOADV
#undef STOREV
}

}  // namespace

namespace aspns {

hipError_t launch_ns_frame4(bool io16, float* state, int32_t* hist, const NsTables* T,
                            const float* in, float* out, int num_streams, hipStream_t s) {
  const dim3 grid((num_streams + 3) / 4), block(256);
  if (io16)
    hipLaunchKernelGGL(ns_frame4_kernel<true>, grid, block, 0, s, state, hist, T, in, out, num_streams);
  else
    hipLaunchKernelGGL(ns_frame4_kernel<false>, grid, block, 0, s, state, hist, T, in, out, num_streams);
  return hipGetLastError();
}

}  // namespace aspns
