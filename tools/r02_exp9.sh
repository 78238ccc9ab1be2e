#!/bin/bash
# sensitivity of the step time to VALU instruction count: N redundant log evaluations (about 60 VALU each) per wave
export TMPDIR=/tmp
O=gpurun_out/r02_exp9; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 400 --warmup 100 --streams-per-wave 3"
run() { python3 bench.py $B "$@" 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
for E in 0 2 4 8; do
  if [ $E = 0 ]; then export ASP_HIPCC_EXTRA=""; else export ASP_HIPCC_EXTRA="ns_kernels1.hip:-DNS1_EXP_EXTRA=$E"; fi
  touch audiosignalprocess_amd/csrc/ns_kernels1.hip
  python -c "from audiosignalprocess_amd import build; build.build_library()"
  echo "== extra $E: split 2"; run --split 2; run --split 2
  echo "== extra $E: split 1"; run --split 1
  echo "== extra $E: S=1024 split 1"; run --split 1 --streams-per-gpu 1024
done
