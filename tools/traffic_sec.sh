#!/bin/bash
# HBM-side traffic of the AEC and BT kernels from PMC counters (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes), at the bench size.  ASP_*_CHAINS=1: one launch per step, so a
# launch's counters are a step's.
export TMPDIR=/tmp ASP_AEC_CHAINS=1 ASP_BT_CHAINS=1
OUT=gpurun_out/traffic_sec; mkdir -p $OUT
for W in aec bt1024; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${W}_$C -- python3 bench.py --workload $W --no-cpu-baseline > $OUT/${W}_$C.json 2> $OUT/${W}_$C.err || echo fail $W $C
  done
done
python3 - <<PY
import csv,glob
for W,kern in (('aec','aec_process_kernel'),('bt1024','bt_macroblock8_kernel')):
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob('$OUT/%s_%s/*/*counter_collection.csv'%(W,C)):
            v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if kern in r['Kernel_Name'] and r['Counter_Name']==C]
            t=v[len(v)//2:]; print(W,C,'launches',len(v),'per-launch (KB) %.6g'%(sum(t)/len(t)), 'per-stream (KB) %.4f'%(sum(t)/len(t)/4096))
PY
