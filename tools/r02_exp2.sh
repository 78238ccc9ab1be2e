#!/bin/bash
# parity of the layout change + the new one-stream-per-wave pair-layout kernel, then its timing
export TMPDIR=/tmp
O=gpurun_out/r02_exp2; mkdir -p $O
python -m pytest tests/test_ns_gpu.py -x -q -m gpu > $O/pytest_ns.txt 2>&1; tail -15 $O/pytest_ns.txt
B="--no-cpu-baseline --no-secondary --no-graph --steps 400 --warmup 100"
for a in "--streams-per-wave 3 --split 1" "--streams-per-wave 3 --split 2" "--streams-per-wave 3 --split 3" "--streams-per-wave 3 --split 4" \
         "--streams-per-wave 3 --split 1 --streams-per-gpu 256" "--streams-per-wave 3 --split 1 --streams-per-gpu 1024" "--streams-per-wave 3 --split 1 --streams-per-gpu 2048" \
         "--streams-per-wave 3 --split 2 --streams-per-gpu 8192" "--streams-per-wave 3 --split 4 --streams-per-gpu 8192" "--streams-per-wave 3 --split 2 --streams-per-gpu 16384" \
         "--streams-per-wave 2 --split 2"; do
  echo "== $a"; timeout -k 10 120 python3 bench.py $B $a 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'], 'min %.2f p90 %.2f wall %.2f' % (1000*d['timing']['ms_per_step_min'], 1000*d['timing']['ms_per_step_p90'], 1000*d['timing']['wall_ms_per_step_median']))"
done
