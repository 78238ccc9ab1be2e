#!/bin/bash
# where does the large-batch (throughput-bound) NS step spend its time?  unit-busy counters, 16384 streams
export TMPDIR=/tmp
K=${1:-3}
OUT=gpurun_out/r02_pmc2_k$K; mkdir -p $OUT
ARGS="--steps 30 --warmup 60 --regions 1 --no-secondary --no-cpu-baseline --streams-per-wave $K --split 1 --streams-per-gpu 16384"
i=0
for set in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "GRBM_TC_BUSY GRBM_EA_BUSY" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed: $set"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$OUT/p*/'), key=lambda s:int(s.rstrip('/').split('p')[-1])):
  for f in glob.glob(d+'/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ns_frame' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        t=v[-30:]; print(k, 'per-launch %.5g'%(sum(t)/len(t)))
  for f in glob.glob(d+'/*/*kernel_trace.csv'):
    rows=[r for r in csv.DictReader(open(f)) if 'ns_frame' in r['Kernel_Name']][-30:]
    if rows: print('  avg kernel us %.2f' % (sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)/len(rows)/1e3))
PY
