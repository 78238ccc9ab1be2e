#!/bin/bash
# one PMC pass: VALU instruction count + wave cycles per launch of the fused kernel
export TMPDIR=/tmp
OUT=gpurun_out/qpmc_$1; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 bench.py --steps 40 --warmup 260 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.err
python3 - <<PY
import csv,glob,collections
for f in glob.glob('$OUT/p1/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ns_frame_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    w=sum(acc['SQ_WAVES'][-40:])/40
    for k,v in sorted(acc.items()):
        t=v[-40:]; print(k, 'per-launch %.4g  per-wave %.1f'%(sum(t)/len(t), sum(t)/len(t)/w))
PY
