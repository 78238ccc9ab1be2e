#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp12; mkdir -p $O
python -m pytest tests/test_ns_gpu.py -q -x 2>&1 | tail -2
B="--no-cpu-baseline --no-secondary --steps 400 --warmup 100"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
for i in 1 2 3; do echo "== k3 split 2"; run --streams-per-wave 3 --split 2; done
echo "== k4 split 2"; run --streams-per-wave 4 --split 2
echo "== k2 split 2"; run --streams-per-wave 2 --split 2
