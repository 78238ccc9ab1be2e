#!/bin/bash
# A/B builds of one kernel file in ONE GPU session: each variant recompiles the named source with
# extra -D / -mllvm flags (audiosignalprocess_amd/build.py reads ASP_HIPCC_EXTRA), runs the dual-kernel
# parity tests and three bench lines.  Used for every "tried and measured" row of profiles/README.md.
#   usage: tools/ab.sh <file.hip> "<flags of variant 1>" ["<flags of variant 2>" ...]
#   e.g.   gpurun -- 'bash tools/ab.sh ns_kernels2.hip "" "-DNS_EXP_FOO=1"'
set -e
FILE=$1; shift
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$TAG]', 'step_us %.2f' % (1000*d['ms_per_step']), '%.2f M/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'])"; }
for FLAGS in "$@"; do
  TAG="${FLAGS:-base}"
  export ASP_HIPCC_EXTRA="$FILE:$FLAGS"
  touch audiosignalprocess_amd/csrc/$FILE
  python -c "from audiosignalprocess_amd import build; build.build_library()"
  python -m pytest tests/test_ns_gpu.py -q -x -k "dual or policies" 2>&1 | tail -1
  run; run; run
done
