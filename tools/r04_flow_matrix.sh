#!/bin/bash
# hand-off build: step time against the steps per timed region, the ring of resident input frames and the steps per launch
B="--no-cpu-baseline --no-secondary --warmup 5"
one() { # label, env/args...
  local lab=$1; shift
  env "${ENVV[@]}" python3 bench.py $B "$@" 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lab step_us %.2f (min %.2f) frac %.3f' % (1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"
}
for ring in 20 100; do
  for K in 20 64 200; do
    for F in on off; do ENVV=(A=1); one "ring $ring K $K flow $F      :" --steps $K --ring $ring --flow $F; done
    for M in 8 16 32; do ENVV=(ASP_NS_FLOW_MAX=$M); one "ring $ring K $K flow on max $M:" --steps $K --ring $ring --flow on; done
  done
done
