#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_default_bench; mkdir -p $O
T0=$(date +%s); python3 bench.py > $O/out.json 2> $O/err.txt; echo "elapsed $(( $(date +%s) - T0 )) s"
python3 -c "
import json; d=json.load(open('$O/out.json')); print('default: steps %d step_us %.2f frac %.3f; secondary %s' % (d['steps'], 1000*d['ms_per_step'], d['roofline']['frac'], [round(1000*s['ms_per_step'],1) for s in d['secondary']]))"
