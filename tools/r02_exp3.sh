#!/bin/bash
# why do 3-4 concurrent chains lose?  kernel-trace timelines + HW-queue count
export TMPDIR=/tmp
O=gpurun_out/r02_exp3; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-graph --steps 400 --warmup 100 --streams-per-wave 3"
for q in 4 8; do
  for sp in 2 4; do
    echo "== GPU_MAX_HW_QUEUES=$q split $sp"; GPU_MAX_HW_QUEUES=$q python3 bench.py $B --split $sp 2>> $O/var.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us %.2f' % (1000*d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"
  done
done
for sp in 2 4; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$sp -- python3 bench.py --no-cpu-baseline --no-secondary --no-graph --steps 40 --warmup 20 --regions 1 --streams-per-wave 3 --split $sp > $O/tr$sp.json 2> $O/tr$sp.err
done
python3 - <<PY
import csv,glob
for sp in (2,4):
    f=glob.glob('$O/tr%d/*/*kernel_trace.csv'%sp)[0]
    rows=[r for r in csv.DictReader(open(f)) if 'ns_frame1' in r['Kernel_Name']]
    rows=rows[-40*sp:]
    t0=min(int(r['Start_Timestamp']) for r in rows)
    print('split',sp,'cols',list(rows[0].keys())[:12])
    for r in rows[:4*sp]:
        print(r.get('Queue_Id'), r.get('Stream_Id'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, r['Grid_Size'] if 'Grid_Size' in r else '')
PY
