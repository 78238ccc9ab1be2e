#!/bin/bash
export TMPDIR=/tmp
echo "tests: $(python -m pytest tests/test_bt_gpu.py -q -x 2>&1 | tail -1)"
for c in 1 2; do for i in 1 2; do ASP_BT_CHAINS=$c python3 bench.py --workload bt1024 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('chains $c: step_us %.1f  %.2f M macroblocks/s frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))"; done; done
