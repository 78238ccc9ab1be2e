#!/bin/bash
# PMC passes of the NS step: the hand-off build (one launch = 20 steps) beside the plain build (one launch per
# step, one chain); per-wave figures.   usage: tools/pmc_ns_flow.sh <outdir> [streams]
export TMPDIR=/tmp
O=${1:-gpurun_out/r04/pmc_flow}; S=${2:-4096}; mkdir -p $O
ARGS="--steps 20 --warmup 260 --no-cpu-baseline --no-secondary --split 1 --regions 3 --streams-per-gpu $S"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM"; do
  i=$((i+1))
  for F in 0 1; do
    ASP_NS_FLOW=$F rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/f${F}p$i -- python3 bench.py $ARGS > $O/f${F}p$i.json 2> $O/f${F}p$i.err || echo "pmc pass $i flow $F failed"
  done
done
python3 - <<PY
import csv,glob,collections
for F in (0,1):
    print("== ASP_NS_FLOW=%d (%s), $S streams: per wave" % (F, "hand-off build, 20 steps per launch" if F else "plain build, one launch per step"))
    for d in sorted(glob.glob('$O/f%dp*/' % F)):
        for f in glob.glob(d+'/*/*counter_collection.csv'):
            acc=collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'ns_frame1_kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
            n = 3 if F else 60       # the timed launches are the last ones
            w=sum(acc['SQ_WAVES'][-n:])/n if acc.get('SQ_WAVES') else 1.0
            for k,v in sorted(acc.items()):
                t=v[-n:]; print('  %-30s per-launch %.5g  per-wave %.1f' % (k, sum(t)/len(t), sum(t)/len(t)/w))
PY
