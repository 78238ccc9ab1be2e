#!/bin/bash
# A/B of aec_kernels.hip builds on the box: each argument is a flag set for that file
export TMPDIR=/tmp
O=gpurun_out/aec_ab; mkdir -p $O
for FLAGS in "$@"; do
  export ASP_HIPCC_EXTRA="aec_kernels.hip:$FLAGS"
  touch audiosignalprocess_amd/csrc/aec_kernels.hip
  python -c "from audiosignalprocess_amd import build; build.build_library()" > $O/build.log 2>&1 || tail -5 $O/build.log
  echo "== $FLAGS: $(python -m pytest tests/test_aec_gpu.py -q -x 2>&1 | tail -1)"
  for i in 1 2; do python3 bench.py --workload aec --no-cpu-baseline 2>> $O/err.txt | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   aec: step_us %.1f  %.2f M frames/s frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))"; done
  python3 tools/aec_stamps.py 2>&1 | tail -2
done
