#!/bin/bash
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --steps 200 --warmup 50 --streams-per-gpu 8192"
for a in "" "--streams-per-wave 3 --split 2" "--streams-per-wave 3 --split 4" "--streams-per-wave 3 --split 1" "--streams-per-wave 2 --split 4" "--streams-per-wave 2 --split 1"; do for i in 1 2; do python3 bench.py $B $a 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[$a] step_us %.2f frac %.3f' % (1000*d['ms_per_step'], d['roofline']['frac']))"; done; done
