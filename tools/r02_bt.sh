#!/bin/bash
# BlockThresholding iteration: parity tests, bench line, phase stamps
export TMPDIR=/tmp
O=gpurun_out/r02_bt; mkdir -p $O
python -m pytest tests/test_bt_gpu.py -q -x > $O/pytest_bt.txt 2>&1; tail -15 $O/pytest_bt.txt
python3 bench.py --workload bt1024 --no-cpu-baseline > $O/bt1024.json 2> $O/bt1024.err; python3 -c "
import json; d=json.load(open('$O/bt1024.json')); print('bt1024: step_us %.1f  %.2f M macroblocks/s frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))" || tail -5 $O/bt1024.err
python3 tools/bt_stamps.py 2>&1 | tail -3
