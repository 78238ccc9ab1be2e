#!/bin/bash
# end-of-round evidence: full GPU suite, smoke, the driver's bench command, NS kernel-trace stats, then the secondary set (tools/r02_profiles.sh)
export TMPDIR=/tmp
TAG=${1:-v6}
O=gpurun_out/r02_final_$TAG; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1 || { tail -20 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -3 $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_${TAG}_bench_steps20.json 2> $O/bench_s20.err || { tail -20 $O/bench_s20.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r02_${TAG}_bench_steps20.json')); print('s20: step_us %.2f value %.1fM frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))
for s in d.get('secondary', []): print(s.get('metric'), s.get('value'), s.get('ms_per_step'), s.get('roofline',{}).get('frac'), s.get('error'))
"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ns_stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/ns_stats.json 2> $O/ns_stats.err && \
  { f=$(find $O/ns_stats -name '*kernel_stats.csv' | head -1); cp $f $O/r02_${TAG}_ns_kernel_stats.csv; head -3 $f | cut -c1-220; } && \
./tools/r02_profiles.sh $TAG
