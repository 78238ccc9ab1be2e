#!/bin/bash
# A/B of bt_kernels8.hip builds on the box: each argument is a flag set for that file
export TMPDIR=/tmp
O=gpurun_out/bt_ab; mkdir -p $O
for FLAGS in "$@"; do
  export ASP_HIPCC_EXTRA="bt_kernels8.hip:$FLAGS"
  touch audiosignalprocess_amd/csrc/bt_kernels8.hip
  python -c "from audiosignalprocess_amd import build; build.build_library()" > $O/build.log 2>&1 || tail -5 $O/build.log
  echo "== $FLAGS: $(python -m pytest tests/test_bt_gpu.py -q -x 2>&1 | tail -1)"
  for i in 1 2; do python3 bench.py --workload bt1024 --no-cpu-baseline 2>> $O/err.txt | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   bt1024: step_us %.1f  %.2f M macroblocks/s frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))"; done
  python3 tools/bt_stamps.py 2>&1 | tail -1
done
