#!/bin/bash
# same-box A / B of the in-tree library against tools/probe/bin/libasp_prev.so (the previous aec_kernels.hip), AEC lines
mkdir -p gpurun_out/r04; O=gpurun_out/r04
[ -n "$AB_NOTESTS" ] || { timeout -k 10 900 python3 -m pytest tests/test_aec_gpu.py -x -q > $O/aec_tests.log 2>&1 || { tail -40 $O/aec_tests.log; exit 1; }; tail -1 $O/aec_tests.log; }
for rep in 1 2 3; do
  for L in "" $PWD/tools/probe/bin/libasp_prev.so; do
    for X in "" "--aec-extended"; do
      ASP_AMD_LIB=$L timeout -k 10 300 python3 bench.py --workload aec --no-cpu-baseline $X 2>$O/aec_bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lib %-5s $X: step_us %.2f frac %.3f' % ('prev' if '$L' else 'build', 1000*d['ms_per_step'], d['roofline']['frac']))" || tail -5 $O/aec_bench.err
    done
  done
done
