#!/bin/bash
# session-2 baseline: full GPU suite, smoke, driver-style bench, NS kernel-trace stats, BT PMC + stamps, AEC stamps
export TMPDIR=/tmp
O=gpurun_out/r02_s2_base; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err
python3 -c "
import json; d=json.load(open('$O/bench_s20.json')); print('s20: step_us %.2f value %.1fM frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))
for s in d.get('secondary', []): print(s.get('metric'), s.get('value'), s.get('ms_per_step'), s.get('roofline',{}).get('frac'), s.get('error'))
"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ns_stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/ns_stats.json 2> $O/ns_stats.err
find $O/ns_stats -name '*kernel_stats.csv' -exec head -5 {} \;
./tools/pmc_bt.sh post_v2 2>&1 | tail -20
python3 tools/bt_stamps.py 2>&1 | tail -3
python3 tools/aec_stamps.py 2>&1 | tail -6
