#!/bin/bash
# round-2 baseline: the driver's command, the long run, the counter list, a VALU-type PMC pass
export TMPDIR=/tmp
O=gpurun_out/r02_base; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/s20.json 2> $O/s20.err; cut -c1-1500 $O/s20.json
B="--no-cpu-baseline --no-secondary"
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-graph" "--steps 1000 --warmup 250" "--steps 1000 --warmup 250 --no-graph" "--steps 200 --warmup 50 --split 1" "--steps 200 --warmup 50 --split 4"; do
  echo "== $a"; python3 bench.py $B $a 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'], d['timing'], d['roofline'].get('copy_ceiling'))"
done
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_INSTS_VALU[A-Z0-9_]*" $O/counters.txt | sort -u | tr '\n' ' '
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 bench.py --steps 40 --warmup 260 --regions 1 --no-graph --no-secondary --no-cpu-baseline > $O/p$i.json 2> $O/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$O/p*/')):
  for f in glob.glob(d+'/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ns_frame' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not acc: continue
    w=sum(acc['SQ_WAVES'][-40:])/40
    for k,v in sorted(acc.items()):
        t=v[-40:]; print(k, 'per-launch %.4g  per-wave %.1f'%(sum(t)/len(t), sum(t)/len(t)/w))
PY
