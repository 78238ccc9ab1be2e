#!/bin/bash
# in-kernel phase stagger sweep (pair-layout kernel)
export TMPDIR=/tmp
O=gpurun_out/r02_exp5; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-graph --steps 400 --warmup 100 --streams-per-wave 3"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
ASP_NS_STAGGER=3000 python -m pytest tests/test_ns_gpu.py -q -x -k "pair" 2>&1 | tail -1
for sg in 0 1000 2000 3000 4000 5000 7000; do
  for sp in 1 2; do
    echo "== stagger $sg split $sp"; ASP_NS_STAGGER=$sg run --split $sp
  done
done
for sg in 2000 4000; do echo "== stagger $sg split 1 S=8192"; ASP_NS_STAGGER=$sg run --split 1 --streams-per-gpu 8192; echo "== stagger $sg split 2 S=8192"; ASP_NS_STAGGER=$sg run --split 2 --streams-per-gpu 8192; done
