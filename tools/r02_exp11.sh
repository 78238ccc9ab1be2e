#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp11; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 400 --warmup 100"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
python -m pytest tests/test_ns_gpu.py -q -x -k "pair" 2>&1 | tail -1
for a in "--split 2" "--split 1" "--split 2 --streams-per-gpu 8192" "--split 1 --streams-per-gpu 1024" "--split 1 --streams-per-gpu 2048"; do
  echo "== kernel 4 $a"; run --streams-per-wave 4 $a
done
echo "== kernel 3 split 2"; run --streams-per-wave 3 --split 2
