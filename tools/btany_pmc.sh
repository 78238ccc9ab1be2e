#!/bin/bash
export TMPDIR=/tmp ASP_BT_CHAINS=1
W=${1:-960}
OUT=gpurun_out/pmc_btany_$W; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 tools/btany_one.py $W > $OUT/p1.txt 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -- python3 tools/btany_one.py $W > $OUT/p2.txt 2> $OUT/p2.err
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$OUT/p*/')):
  for f in glob.glob(d+'/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'bt_macroblock_any' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        t=v[-6:]; print(k, 'per-launch %.4g  per-workgroup %.1f'%(sum(t)/len(t), sum(t)/len(t)/4096))
PY
