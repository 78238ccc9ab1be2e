#!/bin/bash
# A/B of ns_kernels1.hip builds in ONE GPU session: each argument is a flag set for that file
# (audiosignalprocess_amd/build.py reads ASP_HIPCC_EXTRA); per variant the pair-kernel parity tests and
# three bench lines.   e.g. gpurun -- 'bash tools/ns_ab.sh "" "-DNS1_NT_STORE=1"'
export TMPDIR=/tmp
O=gpurun_out/ns_ab; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps ${AB_STEPS:-400} --warmup 100"
for FLAGS in "$@"; do
  export ASP_HIPCC_EXTRA="ns_kernels1.hip:$FLAGS"
  touch audiosignalprocess_amd/csrc/ns_kernels1.hip
  python3 -c "from audiosignalprocess_amd import build; build.build_library()" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
  echo "== '$FLAGS': $(python3 -m pytest tests/test_ns_gpu.py -q -x -k 'pair_kernel_free or pair_kernel_edge' 2>&1 | tail -1)"
  for i in 1 2 3; do python3 bench.py $B 2>> $O/err.txt | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   step_us %.2f (min %.2f) frac %.3f' % (1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"; done
done
