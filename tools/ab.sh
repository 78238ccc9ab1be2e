set -e
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r01_v5_bench.json 2> gpurun_out/r01_v5_bench.err; cat gpurun_out/r01_v5_bench.json
python bench.py --workload aec > gpurun_out/r01_aec_v3_bench.json 2>/dev/null; cat gpurun_out/r01_aec_v3_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v5 -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_v5.json 2> gpurun_out/prof_v5.err
ls gpurun_out/prof_v5/*/ | head
