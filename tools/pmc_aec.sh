#!/bin/bash
# PMC passes for the echo-canceller kernels (separate --pmc runs, kernel trace only).
export TMPDIR=/tmp ASP_AEC_CHAINS=1 ASP_BT_CHAINS=1   # one launch per step: a launch's counters are a step's
OUT=gpurun_out/pmc_aec; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- python3 bench.py --workload aec --steps 80 --warmup 160 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/p2 -- python3 bench.py --workload aec --steps 80 --warmup 160 --no-cpu-baseline > $OUT/p2.json 2> $OUT/p2.err
python3 - <<PY
import csv,glob,collections
for pas in ('p1','p2'):
    for f in glob.glob('$OUT/%s/*/*counter_collection.csv' % pas):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'aec_process_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        n=len(next(iter(acc.values()))) if acc else 0
        print(pas, 'launches', n)
        for k,v in sorted(acc.items()):
            t=v[-20:]; print('  %-24s per-launch %.4g  per-wave(4096) %.1f'%(k, sum(t)/len(t), sum(t)/len(t)/4096))
PY
