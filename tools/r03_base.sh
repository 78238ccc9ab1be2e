#!/bin/bash
# round-3 baseline: bench line, stamps of the default kernel
set -u
mkdir -p gpurun_out/r03_base
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > gpurun_out/r03_base/bench.json 2> gpurun_out/r03_base/bench.err
python3 tools/ns_stamps.py 0 > gpurun_out/r03_base/stamps.txt 2>&1
python3 tools/ns_stamps.py 0 >> gpurun_out/r03_base/stamps.txt 2>&1
cat gpurun_out/r03_base/bench.json gpurun_out/r03_base/stamps.txt
