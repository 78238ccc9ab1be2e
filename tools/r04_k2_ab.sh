#!/bin/bash
# round 4: the two-streams-per-wave kernel in the hand-off build against the pair-layout kernel, same session
mkdir -p gpurun_out/r04; O=gpurun_out/r04
timeout -k 10 900 python3 -m pytest tests/test_ns_gpu.py -x -q -k "handoff or timed_steps or pair_kernel or pair_then" > $O/k2_tests.log 2>&1 || { tail -30 $O/k2_tests.log; exit 1; }
tail -2 $O/k2_tests.log
B="--no-cpu-baseline --no-secondary --warmup 5"
for rep in 1 2; do
  for S in 4096 8192; do
    for K in 3 2; do
      for ST in 20 ${AB_LONG:-200}; do
        python3 bench.py $B --steps $ST --kernel $K --streams-per-gpu $S 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('kernel $K S %d steps %3d: step_us %.2f (min %.2f) frac %.3f' % ($S, $ST, 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"
      done
    done
  done
done
