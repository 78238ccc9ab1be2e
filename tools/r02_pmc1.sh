#!/bin/bash
# PMC passes over the pair-layout NS kernel, one launch per step (counters only)
export TMPDIR=/tmp
TAG=${1:-k3}; shift
OUT=gpurun_out/r02_pmc_$TAG; mkdir -p $OUT
ARGS="--steps 40 --warmup 260 --regions 1 --no-graph --no-secondary --no-cpu-baseline $@"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$OUT/p*/')):
  for f in glob.glob(d+'/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ns_frame' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc: continue
    w=sum(acc['SQ_WAVES'][-40:])/40 if 'SQ_WAVES' in acc else 1
    for k,v in sorted(acc.items()):
        t=v[-40:]; print(k, 'per-launch %.4g  per-wave %.1f'%(sum(t)/len(t), sum(t)/len(t)/w))
  for f in glob.glob(d+'/*/*kernel_trace.csv'):
    rows=[r for r in csv.DictReader(open(f)) if 'ns_frame' in r['Kernel_Name']][-40:]
    if rows: print('  avg kernel us', sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)/len(rows)/1e3)
PY
