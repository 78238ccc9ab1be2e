set -e
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r01_v6_bench.json 2> gpurun_out/r01_v6_bench.err; cat gpurun_out/r01_v6_bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v6 -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_v6.json 2> gpurun_out/prof_v6.err
python tools/ns_stamps.py > gpurun_out/ns_stamps_v6.txt 2>&1; tail -1 gpurun_out/ns_stamps_v6.txt
bash tools/quick_pmc.sh v6 > gpurun_out/qpmc_v6.txt 2>&1; grep -c per-wave gpurun_out/qpmc_v6.txt
bash tools/traffic_ns.sh v6 2>&1 | tail -4
