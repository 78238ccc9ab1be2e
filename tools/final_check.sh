#!/bin/bash
# End-of-round check on the GPU box: full GPU suite, smoke, the three bench lines, rocprofv3 stats of
# the headline command, PMC instruction mix.  Outputs under gpurun_out/final/.
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
python bench.py > $O/ns_bench.json 2> $O/ns_bench.err; cut -c1-250 $O/ns_bench.json
python bench.py --workload aec > $O/aec_bench.json 2> $O/aec_bench.err; cut -c1-200 $O/aec_bench.json
python bench.py --workload bt1024 > $O/bt_bench.json 2> $O/bt_bench.err; cut -c1-200 $O/bt_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof.json 2> $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_aec -- python3 bench.py --workload aec --no-cpu-baseline > $O/prof_aec.json 2> $O/prof_aec.err
bash tools/quick_pmc.sh final > $O/pmc.txt 2>&1
python tools/ns_stamps.py > $O/stamps.txt 2>&1; tail -1 $O/stamps.txt
