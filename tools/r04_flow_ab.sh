#!/bin/bash
# round 4: the hand-off build of the NS step against the launch chains, same box, same session
# (r03 = the library as of the end of round 3, built from that commit into tools/probe/bin/)
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 900 python3 -m pytest tests/test_ns_gpu.py -x -q -k "handoff or timed_steps" > $O/flow_tests.log 2>&1 || { tail -30 $O/flow_tests.log; exit 1; }
tail -3 $O/flow_tests.log
run() {  # tag, env...
  local tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_$tag.json 2> $O/bench_$tag.err || { tail $O/bench_$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag: step_us %.2f frac %.3f config5 %.3f' % (1000*d['ms_per_step'], d['roofline']['frac'], d['config5_single_gpu']['roofline_frac']))"
}
for rep in 1 2; do
  [ -f tools/probe/bin/libasp_r03.so ] && run r03 ASP_AMD_LIB=$PWD/tools/probe/bin/libasp_r03.so ASP_NS_FLOW=0
  run chains ASP_NS_FLOW=0
  run flow ASP_NS_FLOW=1
done
