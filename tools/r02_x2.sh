#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_x2; mkdir -p $O
python3 tools/btany_stamps.py > $O/stamps.txt 2>&1; tail -5 $O/stamps.txt
