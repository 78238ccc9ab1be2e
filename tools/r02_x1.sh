#!/bin/bash
# BT parity (incl. the wide-segment signals) + NS phase stamps of the one-stream-per-wave kernel
export TMPDIR=/tmp
O=gpurun_out/r02_x1; mkdir -p $O
python -m pytest tests/test_bt_gpu.py -q -x > $O/pytest_bt.txt 2>&1; tail -3 $O/pytest_bt.txt
python3 tools/ns_stamps.py 1 > $O/ns_stamps1.txt 2>&1; tail -3 $O/ns_stamps1.txt
