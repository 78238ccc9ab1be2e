#!/bin/bash
# PMC passes over the fused NS kernel (counters only, with --kernel-trace): instruction mix per wave
# and where wave time goes.  usage: tools/quick_pmc.sh <tag> [bench args]
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/qpmc_$TAG; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --steps 40 --warmup 260 --no-cpu-baseline "$@" > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$OUT/p*/')):
  for f in glob.glob(d+'/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ns_frame' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc: continue
    w=sum(acc['SQ_WAVES'][-40:])/40
    for k,v in sorted(acc.items()):
        t=v[-40:]; print(k, 'per-launch %.4g  per-wave %.1f'%(sum(t)/len(t), sum(t)/len(t)/w))
PY
