#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_survey; mkdir -p $O
python3 tools/r02_survey.py > $O/survey.txt 2>&1; tail -8 $O/survey.txt
for W in bt256; do python3 bench.py --workload $W --no-cpu-baseline > $O/$W.json 2> $O/$W.err; python3 -c "
import json; d=json.load(open('$O/$W.json')); print('$W: step_us %.1f value %.3g frac %.3f' % (1000*d['ms_per_step'], d['value'], d['roofline']['frac']))"; done
