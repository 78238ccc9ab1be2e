/*
 * ns_oracle.h -- CPU restatement of the reference's float noise suppressor.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against the
 * reference C (ns_core.c + noise_suppression.c + fft4g.c compiled in place
 * from /root/reference into oracle/_ref/libns_ref.so by oracle/Makefile) in
 * tests/test_ns_oracle_vs_ref.py, and against the committed golden vectors
 * tests/golden/ns_*.npz that the same reference build produced
 * (tests/golden/make_ns_golden.py).
 *
 * State is the canonical AspNsState of include/asp_ns.h.
 */
#ifndef ASP_NS_ORACLE_H_
#define ASP_NS_ORACLE_H_

#include "asp_ns.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How cross-bin sums are associated.
 *   SEQ : left-to-right in bin order, exactly as the reference's loops do
 *         (ns_core.c:951-960, 1088-1101, 538-545, 605-621, 676-683).
 *   TREE: the fixed wave64 butterfly order of the HIP kernels
 *         (slot-local sum, then xor 1,2,4,8,16,32), so the device path can be
 *         checked bit-for-bit against this restatement. */
enum { ASP_NS_REDUCE_SEQ = 0, ASP_NS_REDUCE_TREE = 1, ASP_NS_REDUCE_TREE32 = 2, ASP_NS_REDUCE_TREE64P = 3 };
/*   TREE32: the association of the two-streams-per-wave kernel (ns_kernels2.hip: 32 lanes per stream, four
 *         bins per lane summed in lane order, then xor 1,2,4,8,16, then bin 128). */
/*   TREE64P: the association of the pair-layout kernel (ns_kernels1.hip): the two bins a lane owns,
 *         the wave64 butterfly over the 64 partials, then bin 128. */

int asp_ns_oracle_init(AspNsState* s, uint32_t fs);           /* ns_core.c:74-214   */
int asp_ns_oracle_set_policy(AspNsState* s, int mode);        /* ns_core.c:1013-1041 */
void asp_ns_oracle_analyze(AspNsState* s, const float* frame, /* ns_core.c:1043-1181 */
                           int reduce_mode);
void asp_ns_oracle_process(AspNsState* s, const float* in,    /* ns_core.c:1183-1359 */
                           float* out, int reduce_mode);

/* Batch helper: frames [num_frames][num_streams][160]; Analyze then Process on
 * the same frame per stream (test_ns_module.cpp:97-99). */
/* WebRtcNs_ProcessCore with num_bands = 1 + num_high (ns_core.c:1183-1415): the low band as
 * asp_ns_oracle_process, plus the time-domain gain of the high band(s) (:1227-1235, 1252-1261,
 * 1362-1414).  in_high / out_high [num_high][160]. */
void asp_ns_oracle_process_bands(AspNsState* s, AspNsHbState* hb, const float* in_low,
                                 const float* in_high, int num_high, float* out_low,
                                 float* out_high, int mode);
void asp_ns_oracle_run(AspNsState* states, int num_streams, const float* in,
                       float* out, int num_frames, int reduce_mode);
/* Same, with `threads` pthreads each owning a contiguous shard of streams. */
void asp_ns_oracle_run_mt(AspNsState* states, int num_streams, const float* in,
                          float* out, int num_frames, int reduce_mode,
                          int threads);

/* 256-point real FFT in Ooura packing, in place; WebRtc_rdft(256, isgn, ...)
 * (fft4g.c:324-362).  isgn=+1 forward, -1 inverse (unscaled). */
void asp_ns_oracle_rdft256(float* a, int isgn);
void asp_ns_oracle_rdft128(float* a, int isgn);   /* WebRtc_rdft(128, isgn): the 8 kHz transform */

/* (float)fn((double)x) with the host libm, in place: fn 1 = log, 2 = exp, 3 = tanh. */
void asp_oracle_libm_f32(int fn, float* data, size_t n);

/* Tables shared with the device path (so tests can compare them). */
const float* asp_ns_oracle_window(void);      /* kBlocks160w256, windows_private.h:94-147 */
const float* asp_ns_oracle_fft_w(void);       /* makewt(64) table, fft4g.c:642-669        */
const float* asp_ns_oracle_fft_c(void);       /* makect(64) table, fft4g.c:671-690        */
const float* asp_ns_oracle_window8(void);     /* kBlocks80w128, windows_private.h:64-91   */
const float* asp_ns_oracle_fft_w8(void);      /* makewt(32): WebRtc_rdft(128)             */
const float* asp_ns_oracle_fft_c8(void);      /* makect(32)                               */

#ifdef __cplusplus
}
#endif
#endif
