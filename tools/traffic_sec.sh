#!/bin/bash
# Fabric traffic of the AEC and BT kernels from PMC counters (separate --pmc passes, as MI355X_MICROARCH.md
# prescribes), at the bench size: per stream and frame step (AEC: counter / (Grid_Size / 64) waves, one wave per
# stream and step, the hand-off build's 10-64 steps per launch included) and per macroblock (BT-1024: one workgroup
# of 512 threads per macroblock; BT-256: four macroblocks per workgroup).  ASP_BT_CHAINS=1: one BT launch per step.
export TMPDIR=/tmp ASP_AEC_CHAINS=1 ASP_BT_CHAINS=1
OUT=${1:-gpurun_out/traffic_sec}; mkdir -p $OUT
for W in aec bt1024 bt256; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${W}_$C -- python3 bench.py --workload $W --no-cpu-baseline > $OUT/${W}_$C.json 2> $OUT/${W}_$C.err || echo fail $W $C
  done
done
python3 - <<PY
import csv,glob
for W,kern,units in (('aec','aec_process',lambda g: g/64.0),('bt1024','bt_macroblock8_kernel',lambda g: g/512.0),('bt256','bt_macroblock8_kernel',lambda g: 4*g/512.0)):
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob('$OUT/%s_%s/*/*counter_collection.csv'%(W,C)):
            v=[float(r['Counter_Value'])/units(float(r['Grid_Size'])) for r in csv.DictReader(open(f)) if kern in r['Kernel_Name'] and r['Counter_Name']==C]
            t=v[len(v)//2:]; print(W,C,'launches',len(v),'KB per unit (stream-frame / macroblock) %.4f'%(sum(t)/len(t)))
PY
