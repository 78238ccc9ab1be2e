#!/bin/bash
# A/B of ns_kernels1.hip builds on the box: each argument is a flag set for that file
export TMPDIR=/tmp
O=gpurun_out/r02_ns_ab; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 400 --warmup 100"
for FLAGS in "$@"; do
  export ASP_HIPCC_EXTRA="ns_kernels1.hip:$FLAGS"
  touch audiosignalprocess_amd/csrc/ns_kernels1.hip
  python -c "from audiosignalprocess_amd import build; build.build_library()" > $O/build.log 2>&1 || tail -5 $O/build.log
  echo "== $FLAGS: $(python -m pytest tests/test_ns_gpu.py -q -x -k 'pair' 2>&1 | tail -1)"
  for i in 1 2 3; do python3 bench.py $B 2>> $O/err.txt | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   step_us %.2f frac %.3f' % (1000*d['ms_per_step'], d['roofline']['frac']))"; done
done
