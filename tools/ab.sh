set -e
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r01_v7_bench.json 2> gpurun_out/r01_v7_bench.err; cat gpurun_out/r01_v7_bench.json | cut -c1-260
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v7 -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_v7.json 2> gpurun_out/prof_v7.err
bash tools/quick_pmc.sh v7 > gpurun_out/qpmc_v7.txt 2>&1; grep -c per-wave gpurun_out/qpmc_v7.txt
