#!/bin/bash
# round 4: AEC delay logging in the hand-off build (binary spectra in the process kernel, one estimator launch per
# process launch) against the launch chains (one estimator launch per frame), and against the plain configuration
mkdir -p gpurun_out/r04; O=gpurun_out/r04
[ -n "$AB_NOTESTS" ] || timeout -k 10 900 python3 -m pytest tests/test_aec_gpu.py -x -q -k "logging or handoff or agnostic" > $O/aec_tests.log 2>&1 || { tail -40 $O/aec_tests.log; exit 1; }
[ -n "$AB_NOTESTS" ] || tail -2 $O/aec_tests.log
for rep in 1 2; do
  for F in 1 0; do
    for D in off logging agnostic; do
      ASP_AEC_FLOW=$F timeout -k 10 300 python3 bench.py --workload aec --steps ${AB_STEPS:-1000} --warmup ${AB_WARM:-250} --no-cpu-baseline --aec-delay $D 2>$O/aec_bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('flow $F delay %-8s: step_us %.2f frac %.3f' % ('$D', 1000*d['ms_per_step'], d['roofline']['frac']))" || tail -5 $O/aec_bench.err
    done
  done
done
