#!/bin/bash
# HBM traffic of the fused NS kernel from PMC counters (separate --pmc passes), at the bench
# size (4096 streams, working set inside the 256 MB Infinity Cache) and at 32768 streams
# (1.3 GB of state: every byte must come from / go to HBM), the latter calibrating the counters
# on this kernel's own access pattern as MI355X_MICROARCH.md prescribes.
export TMPDIR=/tmp
OUT=gpurun_out/traffic_$1; mkdir -p $OUT
for S in 4096 32768; do
  R=100; [ $S = 32768 ] && R=12
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${S}_$C -- python3 bench.py --steps 30 --warmup 252 --no-cpu-baseline --split 1 --streams-per-gpu $S --ring $R > $OUT/${S}_$C.json 2> $OUT/${S}_$C.err || echo fail $S $C
  done
done
python3 - <<PY
import csv,glob
for S in (4096,32768):
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob('$OUT/%d_%s/*/*counter_collection.csv'%(S,C)):
            v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if ('ns_frame_kernel' in r['Kernel_Name'] or 'ns_frame2_kernel' in r['Kernel_Name']) and r['Counter_Name']==C]
            t=v[-30:]; print(S,C,'per-launch(raw KB units?) %.6g'%(sum(t)/len(t)), 'per-stream %.4f'%(sum(t)/len(t)/S))
PY
