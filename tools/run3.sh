run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'step_us %.2f' % (1000*d['ms_per_step']), '%.2f M/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'])"; }
python -m pytest tests/test_aec_gpu.py -q -x 2>&1 | tail -1
run --workload aec; run --workload aec
