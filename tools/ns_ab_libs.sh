#!/bin/bash
# same-box A/B of library builds: each argument is a libasp_amd.so path ("" = the in-tree build);
# three bench lines per library, interleaved twice.   e.g. gpurun -- 'bash tools/ns_ab_libs.sh tools/probe/bin/libasp_r02.so ""'
B="--no-cpu-baseline --no-secondary --steps ${AB_STEPS:-400} --warmup 100 ${AB_ARGS:-}"
for rep in 1 2; do
  for L in "$@"; do
    for i in 1 2 3; do ASP_AMD_LIB="$L" python3 bench.py $B 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[%s] step_us %.2f (min %.2f) frac %.3f' % ('${L:-in-tree}', 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"; done
  done
done
