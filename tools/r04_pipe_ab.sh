#!/bin/bash
mkdir -p gpurun_out/r04; O=gpurun_out/r04
timeout -k 10 900 python3 -m pytest tests/test_ns_gpu.py -x -q -k "handoff or timed_steps" > $O/flow_tests.log 2>&1 || { tail -30 $O/flow_tests.log; exit 1; }
tail -1 $O/flow_tests.log
B="--no-cpu-baseline --no-secondary --steps 20 --warmup 5"
for rep in 1 2; do
  for S in 4096 8192; do
    ASP_AMD_LIB=$PWD/tools/probe/bin/libasp_noloop.so python3 bench.py $B --streams-per-gpu $S 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[noloop] S %d step_us %.2f (min %.2f) frac %.3f' % ($S, 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"
    for N in 1 2 4 8; do
      ASP_NS_FLOW_N=$N python3 bench.py $B --streams-per-gpu $S 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[pipe N=$N] S %d step_us %.2f (min %.2f) frac %.3f' % ($S, 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"
    done
  done
done
