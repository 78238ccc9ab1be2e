#!/bin/bash
# round 4: AEC with the next block's FilterFar accumulated during the filter update (AEC_CARRY=1, the build) against
# a library built with -DAEC_CARRY=0 (tools/build_variant.sh carry0 aec_kernels.hip -DAEC_CARRY=0), same box
mkdir -p gpurun_out/r04; O=gpurun_out/r04
[ -n "$AB_NOTESTS" ] || timeout -k 10 900 python3 -m pytest tests/test_aec_gpu.py -x -q > $O/aec_tests.log 2>&1 || { tail -40 $O/aec_tests.log; exit 1; }
[ -n "$AB_NOTESTS" ] || tail -2 $O/aec_tests.log
for rep in 1 2; do
  for L in "" $PWD/tools/probe/bin/libasp_carry0.so; do
    for F in 1 0; do
      for X in "" "--aec-extended"; do
      ASP_AMD_LIB=$L ASP_AEC_FLOW=$F timeout -k 10 300 python3 bench.py --workload aec --steps ${AB_STEPS:-1000} --warmup ${AB_WARM:-250} --no-cpu-baseline $X 2>$O/aec_bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lib %-8s flow $F $X: step_us %.2f frac %.3f' % ('carry0' if '$L' else 'build', 1000*d['ms_per_step'], d['roofline']['frac']))" || tail -5 $O/aec_bench.err
      done
    done
  done
done
