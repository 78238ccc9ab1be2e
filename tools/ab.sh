set -e
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG $*', 'step_us %.2f' % (1000*d['ms_per_step']), '%.2f M/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'])"; }
python -m pytest tests/test_ns_gpu.py -q -x -k "not exhaustive" 2>&1 | tail -1
TAG=f5; run; run; run; run --streams-per-gpu 8192; run --streams-per-gpu 16384
