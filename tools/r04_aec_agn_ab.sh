#!/bin/bash
# round 4: the delay-agnostic mode in one launch per call (control steps and estimator in the stream's own wave)
# against a launch per sub-frame with estimator / control launches between them (ASP_AEC_AGN_FUSED=0), same box
mkdir -p gpurun_out/r04; O=gpurun_out/r04
[ -n "$AB_NOTESTS" ] || timeout -k 10 900 python3 -m pytest tests/test_aec_gpu.py -x -q -k "agnostic or logging or optional or 32" > $O/aec_tests.log 2>&1 || { tail -40 $O/aec_tests.log; exit 1; }
[ -n "$AB_NOTESTS" ] || tail -2 $O/aec_tests.log
for rep in 1 2; do
  for A in 1 0; do
    for X in "" "--aec-extended"; do
      ASP_AEC_AGN_FUSED=$A timeout -k 10 300 python3 bench.py --workload aec --no-cpu-baseline --aec-delay agnostic $X 2>$O/aec_bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fused $A $X: step_us %.2f frac %.3f' % (1000*d['ms_per_step'], d['roofline']['frac']))" || tail -5 $O/aec_bench.err
    done
  done
done
