#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_btany; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_bt_gpu.py -q -x > $O/pytest_bt.txt 2>&1; tail -15 $O/pytest_bt.txt
timeout -k 10 300 python3 tools/r02_btany_perf.py > $O/perf.txt 2>&1; tail -8 $O/perf.txt
