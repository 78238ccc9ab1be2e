// ns_layout.h -- HBM layout of the per-stream noise-suppressor state and of the
// constant tables, shared by the kernels (ns_kernels.hip) and the host side of
// the C-ABI (ns_api.hip).
//
// One stream = one wave64.  Lane q owns spectrum bins q ("slot A") and q+64
// ("slot B"); bin 128 ("slot C") is replicated on every lane and committed by
// lane 0.  Every 129-bin array is stored in natural bin order padded to 132
// floats, so a wave reads it as two fully coalesced 256-byte rows plus one
// broadcast dword.  All arrays of one stream are contiguous (one ~10 KB block
// per stream, 256-byte aligned), fields the fused frame step touches first.
#pragma once
#include <stdint.h>

namespace aspns {

constexpr int kBlockL = 160;   // ns/defines.h:14
constexpr int kAnal = 256;     // ns/defines.h:15
constexpr int kBins = 129;     // ns/defines.h:16
constexpr int kHist = 1000;    // ns/defines.h:45
constexpr int kVecStride = 128;  // one state row = bins 0..127 (bin 128 lives with the scalars): 512 B, cache-line aligned
// Position of bin b inside a state row.  Bins are grouped by (b & 15, b >> 6) = the "dual lane"
// lam = (b & 15) + 16 (b >> 6), which gathers the four bins
// q + 16 t + 64 g (t = 0..3) as one 16-byte access; inside a group the order is t = 0, 2, 1, 3, so
// that lane 2 lam + h of the one-stream-per-wave kernel (ns_kernels1.hip), which owns t = h and
// t = h + 2 (the two outputs of its half of a radix-4 butterfly), moves them as one 8-byte access.
// b < 128.
constexpr int row_pos(int b) {
  return 4 * ((b & 15) + 16 * (b >> 6)) + 2 * ((b >> 4) & 1) + ((b >> 5) & 1);
}
constexpr int kCarry = kAnal - kBlockL;  // 96 live samples of each sliding buffer

// 129-bin arrays, in block order.  "hot" = touched by the fused lock-step step.
enum Vec : int {
  V_LQ0 = 0, V_LQ1, V_LQ2,     // lquantile[3][129]   (ns_core.h:67)
  V_DEN0, V_DEN1, V_DEN2,      // density[3][129]     (ns_core.h:66)
  V_QUANT,                     // quantile            (ns_core.h:68)
  V_SMOOTH,                    // smooth              (ns_core.h:72)
  V_NOISEPREV,                 // noisePrev           (ns_core.h:86)
  V_MAGNPREV_A,                // magnPrevAnalyze     (ns_core.h:88)
  V_LOGLRT,                    // logLrtTimeAvg       (ns_core.h:91)
  V_AVGPAUSE,                  // magnAvgPause        (ns_core.h:95)
  V_HOT_COUNT,
  V_NOISE = V_HOT_COUNT,       // noise               (ns_core.h:85)  cold: == noisePrev while paired
  V_MAGNPREV_P,                // magnPrevProcess     (ns_core.h:90)  cold: == magnPrevAnalyze while paired
  V_INITMAGN,                  // initMagnEst         (ns_core.h:99)  startup only
  V_PARAMNOISE,                // parametricNoise     (ns_core.h:102) startup only
  V_COUNT
};

// per-stream scalars, one dword each (lane k of the wave holds scalar k)
enum Scalar : int {
  S_BLOCKIND = 0, S_UPDATES, S_COUNTER0, S_COUNTER1, S_COUNTER2,
  S_MUP0, S_MUP1, S_MUP2, S_MUP3, S_GAINMAP, S_AGGRMODE, S_INITFLAG,
  S_OVERDRIVE, S_DENOISEBOUND, S_PRIORSPEECHPROB, S_SIGNALENERGY, S_SUMMAGN,
  S_WHITE, S_PINKNUM, S_PINKEXP,
  S_PMP0, S_PMP1, S_PMP2, S_PMP3, S_PMP4, S_PMP5, S_PMP6,
  S_FD0, S_FD1, S_FD2, S_FD3, S_FD4, S_FD5, S_FD6,
  S_FS,
  S_COUNT
};

// Bin 128 of state row f is kept with the scalars, in slot S_TAIL0 + f: it arrives with the two
// scalar loads of a step and leaves with the two scalar stores instead of costing every row a
// one-dword load and a one-dword store of its own.
constexpr int S_TAIL0 = 48;
static_assert(S_COUNT <= S_TAIL0 && S_TAIL0 + V_COUNT <= 64, "row tails share the 64 scalar slots");

// dword offsets inside one stream block
constexpr int kOffScalars = 0;                               // 64 dwords
constexpr int kOffAnaHist = 64;                              // analyzeBuf[160..255]
constexpr int kOffSynt = kOffAnaHist + kCarry;               // syntBuf[0..95]
constexpr int kOffVec = kOffSynt + kCarry;                   // 256
constexpr int kOffDataHist = kOffVec + V_COUNT * kVecStride; // dataBuf[160..255] (cold)
constexpr int kStreamDwordsRaw = kOffDataHist + kCarry;
// dword offset of bin `bin` of state row f inside the stream block
constexpr int row_dword(int f, int bin) {
  return bin < 128 ? kOffVec + f * kVecStride + row_pos(bin) : kOffScalars + S_TAIL0 + f;
}
constexpr int kStreamDwords = (kStreamDwordsRaw + 63) / 64 * 64;  // 256-byte multiple

// histograms: [stream][3][kHistStride] int32 (histLrt, histSpecFlat, histSpecDiff)
constexpr int kHistStride = 1024;
constexpr int kHistDwords = 3 * kHistStride;

// Constant tables (one copy per device).
struct NsTables {
  float window[kAnal];        // kBlocks160w256 (ns/windows_private.h:94-147)
  float tw[3][64][4];         // per pass, per lane: (tAr, tAi, tBr, tBi) -- see ns_kernels.hip
  int32_t diag[64];           // bit s set: pass s uses the w[2] "diagonal" form on this lane
  float cq[64];               // makect table c[q]      (fft4g.c:671-690)
  float cr[64];               // makect table c[64 - q] (lane 0: unused)
  float logi[132];            // (float)log((float)i) for i = 0..128, ns_core.c:1093
  float sum_log_i;            // sequential sums over i = 5..128, ns_core.c:1094-1095
  float sum_log_i_square;
  float pad[2];
  double exp2_64[64];         // 2^(j/64), range-reduction table of the lean exp
  // two-streams-per-wave kernel (ns_kernels2.hip): one full butterfly per lane and pass; that kernel stages
  // tw2 and spl as ONE 4 KB block (256 threads x 16 B), so spl must follow tw2 directly
  float tw2[3][32][8];        // (w1r, w1i, w2r, w2i, w3r, w3i, diag, 0)
  float spl[32][4][2];        // real-split (wkr, wki) of element (lam & 15) + 16 t + 64 (lam >> 4), lam = 0..31
  double logtab[128][2];      // {1/c, log c} of the table-driven log (ns_device.h: log_tab_f64)
  // 8 kHz geometry (ns_kernels.hip, G8): WebRtc_rdft(128) and kBlocks80w128
  float window8[128];         // kBlocks80w128 (ns/windows_private.h:64-91)
  float tw8[2][64][4];        // passes 1 and 2 of the 64-point transform, per lane (lanes 32..63 repeat 0..31)
  int32_t diag8[64];
  float c8a[64];              // makect(32): c[p], p = min(lane, 64 - lane) (0 on lanes 0 and 32)
  float c8b[64];              //             c[32 - p]
  float sum_log_i8;           // sequential sums over i = 5..64, ns_core.c:1094-1095 with magnLen 65
  float sum_log_i_square8;
  float pad8[2];
};

static_assert(__builtin_offsetof(NsTables, logtab) % 16 == 0, "logtab is read as double2");
static_assert(__builtin_offsetof(NsTables, spl) == __builtin_offsetof(NsTables, tw2) + sizeof(float) * 3 * 32 * 8,
              "ns_kernels2.hip stages tw2 and spl as one block");

}  // namespace aspns
