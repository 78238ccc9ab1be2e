#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp6; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-graph --steps 400 --warmup 100 --streams-per-wave 3"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
echo "== base split2"; run --split 2
echo "== base S2048 split1"; run --split 1 --streams-per-gpu 2048
for m in 1 2 3; do for sg in 2000 4000 6000; do
  echo "== mode $m stagger $sg S2048 split 1"; ASP_NS_STAGGER_MODE=$m ASP_NS_STAGGER=$sg run --split 1 --streams-per-gpu 2048
  echo "== mode $m stagger $sg split 2"; ASP_NS_STAGGER_MODE=$m ASP_NS_STAGGER=$sg run --split 2
done; done
