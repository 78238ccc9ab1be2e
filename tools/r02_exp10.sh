#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp10; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 400 --warmup 100"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
for k in 4 3; do
  for a in "--split 2" "--split 1" "--split 3" "--split 2 --streams-per-gpu 8192" "--split 1 --streams-per-gpu 8192" "--split 1 --streams-per-gpu 1024" "--split 1 --streams-per-gpu 2048" "--split 2 --streams-per-gpu 16384"; do
    echo "== kernel $k $a"; run --streams-per-wave $k $a
  done
done
echo "== kernel 4 driver-style"; run --streams-per-wave 4 --split 2 --steps 20 --warmup 5
