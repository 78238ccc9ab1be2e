#!/bin/bash
# PMC passes over the NS bench (counters only with --kernel-trace, as gpurun requires).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
ARGS="--steps 40 --warmup 260 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$OUT/p*/')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'ns_frame_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            tail=v[-40:]
            print(d, k, 'n=%d'%len(v), 'avg_last40=%.4g'%(sum(tail)/len(tail)))
PY
