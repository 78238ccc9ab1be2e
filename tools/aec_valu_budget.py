#!/usr/bin/env python3
"""Per-phase instruction budget of the AEC process kernel's block (costs no GPU time).

Compiles aec_kernels.hip for the device with the build's flags and reads the plain build's
aec_process_kernel<false, 12>: its AEC_STAMP marks survive there as s_memtime instructions (the stamps pointer is a
run-time argument), so the instructions between two marks, in layout order, are the phase's code.  STATIC counts:
cold blocks (the comfort-noise order statistics only with hnl_kind == 2, PartitionDelay every 10th block, the libm
fall-backs of pow / sincos, the non-carried FilterFar of a call's first block) sit inside their phase; nlp_pow,
nlp_sincos, high_band_block and metrics_block are out of line and not counted.  A guide to where the block's
~2 900 executed VALU instructions are, not a cycle count.

usage: tools/aec_valu_budget.py [extra hipcc flags...]
"""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["(before the first block: staging, far-end work)", "loads + far spectra + FilterFar (first block of a call)",
         "near FFT (+ echo estimate's inverse, one mixed round)", "power smoothing, noise floor, xf buffer", "-", "echo estimate, error",
         "error FFT", "ScaleErrorSignal", "FilterAdaptation (3 rounds x 4 partitions, inverse + forward) + carried FilterFar",
         "PartitionDelay", "-", "SmoothedPSD, coherence, the four band sums", "NLP scalars, hNl, order statistics",
         "overdrive, suppression, comfort noise", "inverse FFT, overlap-add", "carry the block, scalars, bin-64 column",
         "(after the block loop)"]


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "lane"
        if op.startswith("v_cmp"):
            return "cmp"
        if op.startswith("v_pk_"):
            return "pk_f32"
        if op.startswith(("v_mov_b", "v_accvgpr")):
            return "mov"
        if op.endswith("_f64") or "_f64_" in op:
            return "f64"
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "aec.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function", "-mllvm",
               "-amdgpu-sched-strategy=iterative-ilp", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "audiosignalprocess_amd", "csrc"), "--cuda-device-only", "-S",
               os.path.join(ROOT, "audiosignalprocess_amd", "csrc", "aec_kernels.hip"), "-o", out] + extra
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "aec_process_kernelILb0ELi12E" in l and l.rstrip().endswith(":") or
                 (l.startswith("_ZN") and "aec_process_kernelILb0ELi12E" in l and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    segs = [collections.Counter()]
    for l in lines[start:end]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_memtime":
            segs.append(collections.Counter())
            continue
        c = classify(op)
        if c:
            segs[-1][c] += 1
    cols = ["pk_f32", "valu_other", "cndmask", "cmp", "mov", "lane", "f64", "lds", "vmem", "salu", "branch", "waitcnt"]
    print("aec_process_kernel<false, 12>, static instruction counts between the phase marks (layout order)")
    print("%-4s %-88s %6s | %s" % ("mark", "phase", "VALU", " ".join("%7s" % c for c in cols)))
    tot = 0
    for k, c in enumerate(segs):
        valu = sum(c[x] for x in ("pk_f32", "valu_other", "cndmask", "cmp", "mov", "lane", "f64"))
        name = NAMES[k] if k < len(NAMES) else "?"
        if 1 <= k <= 15:
            tot += valu
        print("%-4d %-88s %6d | %s" % (k, name, valu, " ".join("%7d" % c[x] for x in cols)))
    print("VALU of one block, marks 1..15 (static): %d; executed per block by the counters (r04_aec_pmc_flow_vs_plain.txt): ~2 900 of which"
          " ~100 each in nlp_pow x 2 and nlp_sincos (out of line)" % tot)


if __name__ == "__main__":
    main()
