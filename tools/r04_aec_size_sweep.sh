#!/bin/bash
# round 4: AEC (hand-off build) over the batch size: frames/s and roofline fraction per streams-per-GPU
mkdir -p gpurun_out/r04; O=gpurun_out/r04
for S in 1024 2048 4096 6144 8192 12288 16384; do
  for X in "" "--aec-extended"; do
    timeout -k 10 300 python3 bench.py --workload aec --no-cpu-baseline --streams-per-gpu $S $X 2>$O/aec_bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('S %5d $X: step_us %.2f  %.1f M frames/s  frac %.3f' % ($S, 1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))" || tail -5 $O/aec_bench.err
  done
done
