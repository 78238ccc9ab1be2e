// aec_layout.h -- HBM layout of the batched echo canceller and the launch descriptors shared by
// aec_kernels.hip (device) and aec_api.hip (host control plane).
//
// Per stream (all float / int32 dwords, one contiguous block of state_dwords(NP)), NP = 12 partitions, or
// 32 with the extended filter (aec_core_internal.h:23-25).  The blocks whose size does not depend on NP
// come first, at the same offsets for both filter lengths:
//   c64[256]: bin 64 of every state row, one contiguous column a wave gathers with 2 (NP = 12) / 4 loads
//   dBuf[128] eBuf[128] outBuf[64]
//   scalars[32]: see S_* below
//   rings: far_pre[448] nearFr[144] outFr[144]   (positions live on the host), the 32 kHz high band's
//   delay line and rings
// then the rows of kRowS = 64 dwords (256-byte aligned) holding bins 0..63 (lane q <-> bin q):
//     xPow dPow dMinPow dInitMinPow sx sd se sde.re sde.im sxd.re sxd.im           (11 rows)
//     xfBuf  re[NP] im[NP]   -- partition p at physical index p (circular, xfBufBlockPos on host)
//     wfBuf  re[NP] im[NP]
//     xfwBuf [32] x {re, im} -- circular, newest at physical index `xfw_head` (host); all 32 blocks of
//                               history whatever NP, as the reference keeps them (aec_core.c:1079-1081):
//                               switching the filter length mid-stream then reads what the reference reads
// Far-end spectra ring: separate allocation [250 slots][stream][4 rows of kRow]:
//   plain re, plain im, windowed re, windowed im  (far_buf / far_buf_windowed, aec_core.c:1330-1341)
#pragma once
#include <stdint.h>

#include "asp_aec.h"  // AspAecDelayState: the per-stream delay-estimator block is the canonical struct itself

namespace aspaec {

constexpr int kPartLen = 64, kPartLen1 = 65, kPartLen2 = 128, kFrameLen = 80;
constexpr int kNumPartNormal = 12, kNumPartMax = 32;  // kNormalNumPartitions / kExtendedNumPartitions
constexpr int kFarSlots = 250;              // kBufSizePartitions, aec_core.c:37
constexpr int kPreLen = 128 + 4 * 80;       // far_pre_buf, echo_cancellation.c:146-147
constexpr int kFrBufLen = 80 + 64;          // nearFrBuf / outFrBuf, aec_core.c:1299-1305
constexpr int kRow = 68;    // far-ring rows: 65 bins + pad
constexpr int kRowS = 64;   // state rows: bins 0..63 (bin 64 in the c64 column)

// rows that exist once, whatever the filter length; the partition rows follow from R_XF_RE on
enum Row {
  R_XPOW = 0, R_DPOW, R_DMINPOW, R_DINITMINPOW, R_SX, R_SD, R_SE, R_SDE_RE, R_SDE_IM, R_SXD_RE,
  R_SXD_IM,
  R_XF_RE                           // NP rows, then R_XF_IM, R_WF_RE, R_WF_IM (NP each), R_XFW (2 x 32)
};
// row indices that depend on the number of partitions
struct AecRows {
  int NP, R_XF_IM, R_WF_RE, R_WF_IM, R_XFW, R_COUNT, state_dwords;
  constexpr explicit AecRows(int np);
};

constexpr int kC64Len = 256;                      // >= R_COUNT of 32 partitions (203)
constexpr int kOffC64 = 0;
constexpr int kOffDBuf = kOffC64 + kC64Len;       // the sample buffers are 256-byte aligned
constexpr int kOffEBuf = kOffDBuf + 128;
constexpr int kOffOutBuf = kOffEBuf + 128;
constexpr int kOffScalars = kOffOutBuf + 64;
constexpr int kOffPre = kOffScalars + 32;
constexpr int kOffNearFr = kOffPre + kPreLen;
constexpr int kOffOutFr = kOffNearFr + kFrBufLen;
// 32 kHz: the high band's delay line and its two rings (positions shared with the low band's)
constexpr int kOffDBufH = kOffOutFr + kFrBufLen;   // dBufH[0][0..63]
constexpr int kOffNearFrH = kOffDBufH + 64;
constexpr int kOffOutFrH = kOffNearFrH + kFrBufLen;
constexpr int kOffRows = ((kOffOutFrH + kFrBufLen + 63) / 64) * 64;

constexpr AecRows::AecRows(int np)
    : NP(np), R_XF_IM(R_XF_RE + np), R_WF_RE(R_XF_RE + 2 * np), R_WF_IM(R_XF_RE + 3 * np),
      R_XFW(R_XF_RE + 4 * np), R_COUNT(R_XF_RE + 4 * np + 2 * kNumPartMax),
      state_dwords(kOffRows + (R_XF_RE + 4 * np + 2 * kNumPartMax) * kRowS) {}
static_assert(AecRows(kNumPartMax).R_COUNT <= kC64Len, "the bin-64 column holds every row of the extended filter");

enum Scalar {
  S_HNLFBMIN = 0, S_HNLFBLOCALMIN, S_HNLXDAVGMIN, S_OVERDRIVE, S_OVERDRIVESM,  // float
  S_HNLNEWMIN, S_HNLMINCTR, S_DELAYIDX, S_STNEARSTATE, S_ECHOSTATE, S_DIVERGESTATE,  // int
  S_NOISEESTCTR, S_DELAYESTCTR, S_SEED
};

constexpr int kFarSlotDwords = 4 * kRow;  // per stream per slot

// Echo metrics (metricsMode): a separate allocation [stream][kMetDwords] whose first 65 dwords are
// the AspAecMetricsState image: 4 PowerLevel x 7, 4 Stats x 9 (erl, erle, aNlp, rerl), stateCounter.
constexpr int kMetLevel = 7, kMetStat = 9, kMetStats = 4 * kMetLevel;
constexpr int kMetStateCounter = kMetStats + 4 * kMetStat;  // 64
constexpr int kMetDwords = 80;

// Constant tables (host-built, aec_api.hip; the kernels stage them in LDS).
struct AecTables {
  float w[64];         // rdft_w, aec_rdft.c:32-49
  float wk3a[16];      // rdft_wk3ri_first
  float wk3b[16];      // rdft_wk3ri_second
  float hann[68];      // WebRtcAec_sqrtHanning[65]
  float weight[68];    // WebRtcAec_weightCurve[65]
  float odrive[68];    // WebRtcAec_overDriveCurve[65]
  uint32_t lcg_a[64];  // 69069^(k+1) mod 2^32
  uint32_t lcg_c[64];  // sum_{i<=k} 69069^i mod 2^32
  double exp2_64[64];  // 2^(j/64) of the lean exp (ns_device.h)
};

// One WebRtcAec_BufferFarend call (echo_cancellation.c:278-339) as the device sees it.
struct FarOps {
  int32_t n;         // samples appended to far_pre (80 / 160); 0: nothing to append
  int32_t wpos;      // write position in far_pre before the append
  int32_t nparts;    // partitions transformed by this call (0..3)
  int32_t rpos[3];   // far_pre read position of each partition's 128 samples
  int32_t slot[3];   // far ring slot each partition is written to
};

struct BlockOp {     // one ProcessBlock (aec_core.c:1084-1287)
  int32_t near_rpos; // nearFrBuf read position of the 64 new samples
  int32_t far_slot;  // far ring slot consumed
  int32_t out_wpos;  // outFrBuf write position
  int32_t xf_pos;    // xfBufBlockPos after its decrement
  int32_t xfw_head;  // physical xfwBuf partition receiving this block's windowed far spectrum
};

struct SubFrame {    // one FRAME_LEN iteration of WebRtcAec_ProcessFrames (aec_core.c:1679-1777)
  int32_t near_wpos;
  int32_t nblocks;   // 0..2
  BlockOp blk[2];
  int32_t out_rpos;  // outFrBuf read position after the stuffing step
};

// One WebRtcAec_Process call in the running (non start-up) phase.
struct ProcOps {
  int32_t nsub;      // 1 or 2 sub-frames of 80 samples
  int32_t mult;      // sampFreq / 8000
  int32_t nlp_mode;
  int32_t num_high;  // 0, or 1 at 32 kHz
  float mu, error_threshold;
  int32_t spectra;   // 1: every block leaves its far / near power spectra in the scratch (delay estimation on)
  int32_t agnostic;  // 1: the blocks' far slots come from the stream's DelayBlock (delay-agnostic mode)
  SubFrame sub[2];
};

// One frame step of a hand-off launch (aec_kernels.hip, AecFlowArgs): the WebRtcAec_BufferFarend call that
// preceded the WebRtcAec_Process call (when its device work fits one descriptor), the Process call, and the
// frames they take and emit.  An array of these in device memory, one per grid row, written before the launch.
struct AecFlowStep {
  ProcOps ops;
  FarOps fops;
  int32_t spec_base;     // delay logging: index of this step's first block in the launch's binary-spectra scratch (-1: off)
  const float* farend;   // nullptr: no far-end work in this step
  const float* nearend;
  float* out;
};
static_assert(sizeof(AecFlowStep) % 8 == 0, "steps stay 8-byte aligned in their array");

// Per-stream control (AspAecBatch_ProcessV): what one stream does in this WebRtcAec_Process call.
struct AecStreamStep {
  int32_t mode;   // 0: pass the near end through (start-up phase, echo_cancellation.c:660-664 / 768-776); 1: `ops`
  int32_t pad;
  ProcOps ops;
};
static_assert(sizeof(AecStreamStep) % 8 == 0, "steps stay 8-byte aligned in their array");

// ---- delay estimation (set_config delay_logging) and the delay-agnostic mode (reported delays off) ----
// Per stream one DelayBlock: the canonical estimator state, then what only the device needs.  In the agnostic
// mode the stream's own far-buffer read side and system delay live in it (s.far_read .. s.system_delay): the
// estimator moves them apart between the streams of a batch, so that part of the control plane runs per stream on
// the device (aec_delay_kernels.hip) and hands the far-ring slots of the coming blocks to the process kernel.
struct DelayBlock {
  AspAecDelayState s;
  int32_t slot[2];   // far-ring slots of the coming sub-frame's blocks (agnostic mode)
  int32_t pad[3];
};
static_assert(sizeof(DelayBlock) % 16 == 0, "delay blocks stay 16-byte aligned");

// skew compensation (aec_resampler.h:16-21, aec_resampler.c:23-25)
constexpr int kResamplingDelay = 1, kResamplerBufferSize = 4 * kFrameLen, kSkewEstimateFrames = 400;

constexpr int kSpecBlocks = 4;             // power spectra of up to 4 blocks wait for the estimator
constexpr int kFlowBitsBlocks = 256;       // hand-off build: binary far / near spectra of up to 64 steps x 4 blocks wait for the estimator
constexpr int kSpecDwords = 2 * kRow;      // per block: |X|^2 then |D|^2, 65 bins each (aec_core.c:1148-1155)
constexpr int kMaxFarEvents = 8;

struct DelayOps {
  int32_t npending;  // blocks of the preceding process launch whose power spectra wait in the scratch
  int32_t logging;   // delay_logging_enabled: estimates go into delay_histogram (aec_core.c:1199-1202)
  int32_t control;   // agnostic mode: run the stream's far-buffer control of the coming sub-frame (aec_core.c:1696-1751)
  int32_t sync;      // first control step after Init: start from the batch-wide values below
  int32_t h_far_read, h_far_write, h_far_wrap, h_system_delay;
  int32_t mult, num_part;
  int32_t nblocks;   // blocks of the coming sub-frame (0..2)
  int32_t nevents;   // WebRtcAec_BufferFarend calls since the last control step: samples and partitions of each
  int32_t ev_samples[kMaxFarEvents], ev_parts[kMaxFarEvents];
};

// The delay-agnostic mode fused into the Process launch (aec_kernels.hip, aec_process_agn_kernel): the control step
// of each 80-sample sub-frame, run by the stream's own wave before the sub-frame's blocks, and the estimator's share
// of those blocks right after them.
struct AgnOps {
  int32_t logging;   // delay_logging_enabled: the estimator runs (aec_core.c:1191-1203)
  int32_t pad;
  DelayOps sub[2];   // control = 1, nblocks, and for the call's first sub-frame the far-end calls to replay / the sync values
};

// A frame step of a hand-off launch in the delay-agnostic mode: the step and its sub-frames' control descriptors
struct AecFlowStepAgn {
  AecFlowStep step;
  AgnOps agn;
};
static_assert(sizeof(AecFlowStepAgn) % 8 == 0, "steps stay 8-byte aligned in their array");

}  // namespace aspaec
