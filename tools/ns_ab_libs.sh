#!/bin/bash
# same-box A/B of library builds: each argument is a libasp_amd.so path ("" = the in-tree build);
# bench lines at 4096 and 8192 streams per library, interleaved twice.
#   e.g. gpurun -- 'bash tools/ns_ab_libs.sh "" tools/probe/bin/libasp_occ5.so'     (AB_STEPS, AB_ARGS, AB_SIZES)
B="--no-cpu-baseline --no-secondary --steps ${AB_STEPS:-20} --warmup 5 ${AB_ARGS:-}"
for rep in 1 2; do
  for L in "$@"; do
    for S in ${AB_SIZES:-4096 8192}; do ASP_AMD_LIB="${L:+$PWD/$L}" python3 bench.py $B --streams-per-gpu $S 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[%s] S %d step_us %.2f (min %.2f) frac %.3f' % ('${L:-in-tree}', $S, 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"; done
  done
done
