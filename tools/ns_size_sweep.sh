#!/bin/bash
# step time of the default kernel against the batch size and the number of launch chains (one box, one session)
for S in ${SWEEP_S:-512 1024 2048 3072 4096 6144}; do
  for split in ${SWEEP_SPLIT:-1 2}; do
    python3 bench.py --no-cpu-baseline --no-secondary --steps 400 --warmup 100 --streams-per-gpu $S --split $split --kernel 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('S %5d split %d: step_us %.2f (min %.2f) frac %.3f' % ($S, $split, 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"
  done
done
