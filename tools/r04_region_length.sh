#!/bin/bash
# round 4: step time against the length of the timed region (fixed cost per region = host recording ahead of the first
# launch, launch-boundary drains, clock ramp), hand-off build and launch chains
mkdir -p gpurun_out/r04; O=gpurun_out/r04
for W in aec bt1024; do
  for F in 1 0; do
    for K in 1000 2000 4000 8000 16000; do
      ASP_AEC_FLOW=$F ASP_BT_FLOW=$F timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline --steps $K 2>$O/len.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$W flow $F: %5d steps  %.2f us per step  region %.2f ms' % (d['steps'], 1000*d['ms_per_step'], d['steps']*d['ms_per_step']))" || tail -3 $O/len.err
    done
  done
done
