#!/bin/bash
# PMC passes for the echo-canceller process kernel, per wave (= per stream and WebRtcAec_Process call):
# the plain build with one launch chain (ASP_AEC_FLOW=0 ASP_AEC_CHAINS=1) beside the hand-off build.
export TMPDIR=/tmp ASP_AEC_CHAINS=1
OUT=${1:-gpurun_out/pmc_aec}; mkdir -p $OUT
for F in 0 1; do
  export ASP_AEC_FLOW=$F
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $OUT/f${F}p1 -- python3 bench.py --workload aec --steps 40 --warmup 160 --no-cpu-baseline > $OUT/f${F}p1.json 2> $OUT/f${F}p1.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/f${F}p2 -- python3 bench.py --workload aec --steps 40 --warmup 160 --no-cpu-baseline > $OUT/f${F}p2.json 2> $OUT/f${F}p2.err
  rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $OUT/f${F}p3 -- python3 bench.py --workload aec --steps 40 --warmup 160 --no-cpu-baseline > $OUT/f${F}p3.json 2> $OUT/f${F}p3.err
done
python3 - <<PY
import csv,glob,collections
for F in (0,1):
  print("== ASP_AEC_FLOW=%d: per wave" % F)
  for pas in ('p1','p2','p3'):
    for f in glob.glob('$OUT/f%d%s/*/*counter_collection.csv' % (F,pas)):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'aec_process' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value'])/(float(r['Grid_Size'])/64.0))
        for k,v in sorted(acc.items()):
            t=v[-(3 if F else 100):]; print('  %-28s %.1f'%(k, sum(t)/len(t)))
PY
