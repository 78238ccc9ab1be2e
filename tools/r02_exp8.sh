#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp8; mkdir -p $O
B="--no-cpu-baseline --no-secondary --streams-per-wave 3 --split 2"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['timing']; print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'], 'min %.2f p10 %.2f p90 %.2f wall %.2f' % (1000*t['ms_per_step_min'],1000*t['ms_per_step_p10'],1000*t['ms_per_step_p90'],1000*t['wall_ms_per_step_median']))"; }
python -m pytest tests/test_ns_gpu.py -q -x -k "pair or dual" 2>&1 | tail -1
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-gate" "--steps 1000 --warmup 250" "--steps 1000 --warmup 250 --no-gate" "--steps 100 --warmup 20" "--steps 5 --warmup 5"; do echo "== $a"; run $a; done
