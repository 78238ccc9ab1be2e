#!/bin/bash
# N = 256 BlockThresholding: bench line + rocprofv3 kernel stats
export TMPDIR=/tmp
O=gpurun_out/r02_bt256; mkdir -p $O
python3 bench.py --workload bt256 > $O/r02_bt256_v1_bench.json 2> $O/bench.err || tail -5 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/r02_bt256_v1_bench.json')); print('bt256: step_us %.1f value %.3g frac %.3f cpu %s' % (1000*d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value')))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload bt256 --no-cpu-baseline > $O/stats.json 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/r02_bt256_v1_kernel_stats.csv; head -4 $f | cut -c1-220
