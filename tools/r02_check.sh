#!/bin/bash
# full GPU suite + the driver's bench command + the default bench
export TMPDIR=/tmp
O=gpurun_out/r02_check; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; python3 -c "
import json; d=json.load(open('$O/bench_s20.json')); print('s20: step_us %.2f value %.1fM frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac'])); print(d['timing']); print({k:v for k,v in d['roofline'].items() if 'ceiling' in k}); print(d.get('config5_single_gpu')); print(d.get('cpu_baseline'))
for s in d.get('secondary', []): print(s.get('metric'), s.get('value'), s.get('ms_per_step'), s.get('roofline',{}).get('frac'), s.get('cpu_baseline'), s.get('error'))
"
python3 bench.py --no-secondary --no-cpu-baseline > $O/bench_s1000.json 2> $O/bench_s1000.err; python3 -c "
import json; d=json.load(open('$O/bench_s1000.json')); print('s1000: step_us %.2f value %.1fM frac %.3f' % (1000*d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))"
