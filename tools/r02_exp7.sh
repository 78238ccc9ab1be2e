#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp7; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-graph --steps 400 --warmup 100 --streams-per-wave 2"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
ASP_NS_STAGGER_MODE=2 ASP_NS_STAGGER=6000 python -m pytest tests/test_ns_gpu.py -q -x -k "dual" 2>&1 | tail -1
echo "== base split1"; run --split 1
echo "== base split2"; run --split 2
for m in 2 3 1; do for sg in 3000 6000 9000 12000; do
  echo "== mode $m stagger $sg split 1"; ASP_NS_STAGGER_MODE=$m ASP_NS_STAGGER=$sg run --split 1
done; done
for sg in 3000 6000 9000; do echo "== mode 1 stagger $sg split 2"; ASP_NS_STAGGER_MODE=1 ASP_NS_STAGGER=$sg run --split 2; done
echo "== k3 base split2 (frontload 0)"; python3 bench.py --no-cpu-baseline --no-secondary --no-graph --steps 400 --warmup 100 --streams-per-wave 3 --split 2 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"
