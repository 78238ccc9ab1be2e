#!/bin/bash
# 48 kHz three-band split / merge: parity tests, bench line, kernel stats
export TMPDIR=/tmp
TAG=${1:-x}
O=gpurun_out/r02_split_$TAG; mkdir -p $O
python -m pytest tests/test_qmf_gpu.py tests/test_sinc_gpu.py tests/test_ns_gpu.py -q -x -m gpu -k "qmf or sinc or split or apm or band or 48 or 32" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 bench.py --workload split48 > $O/r02_split48_${TAG}_bench.json 2> $O/bench.err || tail -5 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/r02_split48_${TAG}_bench.json')); print('split48: step_us %.1f  %.2f M channel-frames/s' % (1000*d['ms_per_step'], d['value']/1e6))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload split48 > $O/stats.json 2> $O/stats.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/r02_split48_${TAG}_kernel_stats.csv; head -8 $f | cut -c1-230
