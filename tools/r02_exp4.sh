#!/bin/bash
# memory floor probe + front-load A/B of the pair-layout kernel
export TMPDIR=/tmp
O=gpurun_out/r02_exp4; mkdir -p $O
./tools/probe/bin/mem_probe > $O/mem_probe.txt 2>&1; cat $O/mem_probe.txt
B="--no-cpu-baseline --no-secondary --no-graph --steps 400 --warmup 100 --streams-per-wave 3"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
for FLAGS in "-DNS1_FRONTLOAD=1" "-DNS1_FRONTLOAD=0"; do
  export ASP_HIPCC_EXTRA="ns_kernels1.hip:$FLAGS"
  touch audiosignalprocess_amd/csrc/ns_kernels1.hip
  python -c "from audiosignalprocess_amd import build; build.build_library()"
  python -m pytest tests/test_ns_gpu.py -q -x -k "pair" 2>&1 | tail -1
  echo "== $FLAGS split 2"; run --split 2; run --split 2
  echo "== $FLAGS split 1"; run --split 1
  echo "== $FLAGS split 4 q8"; GPU_MAX_HW_QUEUES=8 run --split 4
  echo "== $FLAGS split 3 q8"; GPU_MAX_HW_QUEUES=8 run --split 3
  echo "== $FLAGS 8192 split 2"; run --split 2 --streams-per-gpu 8192
done
