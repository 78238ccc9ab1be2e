#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_benchline; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('s20: step_us %.2f frac %.3f' % (1000*d['ms_per_step'], d['roofline']['frac'])); print(d.get('pcie_inclusive'))"
