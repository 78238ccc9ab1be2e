#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_x2; mkdir -p $O
python3 tools/r02_btany_split.py > $O/split.txt 2>&1; tail -5 $O/split.txt
