#!/bin/bash
# Evidence set of the NS headline for profiles/ (one GPU session): the driver's bench line, rocprofv3 kernel
# stats of the same command, PMC instruction mix and wait / busy cycles per wave, fabric traffic (FETCH_SIZE /
# WRITE_SIZE in separate passes, at the bench size and at 32768 streams where every byte is HBM traffic: the
# calibration the guide asks for), phase stamps of a wave in the middle of a hand-off launch, batch-size sweep,
# same-session A / B against the plain build (launch chains).   usage: tools/profiles_ns.sh <tag>   (writes gpurun_out/<tag>/)
# Every figure is per wave = per stream-frame (Grid_Size / 64 waves per launch), so the hand-off build's
# launches (up to 64 frame steps each) and the plain build's (one step each) read alike.
export TMPDIR=/tmp
TAG=${1:-r04}; O=gpurun_out/$TAG; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_steps20.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.json 2> $O/stats.err
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/${TAG}_ns_kernel_stats.csv
ARGS="--steps 40 --warmup 260 --no-cpu-baseline --no-secondary --split 1 --regions 5"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  for F in on off; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p${F}$i -- python3 bench.py $ARGS --flow $F > $O/p${F}$i.json 2> $O/p${F}$i.err || echo "pmc pass $i flow $F failed"
  done
done
for S in 4096 32768; do
  R=100; [ $S = 32768 ] && R=12
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/t${S}_$C -- python3 bench.py --steps 30 --warmup 252 --no-cpu-baseline --no-secondary --split 1 --regions 5 --streams-per-gpu $S --ring $R > $O/t${S}_$C.json 2> $O/t${S}_$C.err || echo fail $S $C
  done
done
python3 - > $O/${TAG}_ns_pmc.txt <<PY
import csv,glob,collections
for F,name in (("on","hand-off build ns_frame1_kernel<false, true>, 40 frame steps per launch"),("off","plain build ns_frame1_kernel<false, false>, one launch per frame step (--split 1)")):
    print("== %s: per wave (= per stream-frame), timed launches only" % name)
    for d in sorted(glob.glob('$O/p%s*/' % F)):
        for f in glob.glob(d+'/*/*counter_collection.csv'):
            acc=collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'ns_frame1_kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value'])/(float(r['Grid_Size'])/64.0))
            n = 5 if F=="on" else 200       # the timed launches are the last ones
            for k,v in sorted(acc.items()):
                t=v[-n:]; print('  %-30s %.1f' % (k, sum(t)/len(t)))
PY
python3 - > $O/${TAG}_ns_traffic.txt <<PY
import csv,glob
print("fabric traffic of the hand-off build ns_frame1_kernel<false, true>, KB per stream-frame (counter / (Grid_Size / 64)), timed launches (30 frame steps each):")
res={}
for S in (4096,32768):
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob('$O/t%d_%s/*/*counter_collection.csv'%(S,C)):
            v=[float(r['Counter_Value'])/(float(r['Grid_Size'])/64.0) for r in csv.DictReader(open(f)) if 'ns_frame1_kernel' in r['Kernel_Name'] and r['Counter_Name']==C]
            t=v[-5:]; res[(S,C)]=sum(t)/len(t); print(S,C,'per stream-frame %.4f' % (sum(t)/len(t)))
print(res)
PY
for N in 1; do timeout -k 10 200 python3 tools/ns_flow_stamps.py 4096 8192; done > $O/${TAG}_ns_flow_stamps.txt 2>&1
{ for S in 1024 2048 4096 8192 16384; do for F in on off; do python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --flow $F --streams-per-gpu $S 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('S %5d flow %-3s step_us %.2f (min %.2f) frac %.3f' % ($S, '$F', 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"; done; done; } > $O/${TAG}_ns_size_sweep.txt 2>&1
{ for K in 2 5 20 64 200 1000; do python3 bench.py --no-cpu-baseline --no-secondary --steps $K --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('steps per timed region %4d: step_us %.2f (min %.2f) frac %.3f' % ($K, 1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"; done; } > $O/${TAG}_ns_steps_sweep.txt 2>&1
tail -3 $O/${TAG}_ns_traffic.txt; head -3 $O/${TAG}_ns_kernel_stats.csv | cut -c1-200; cat $O/${TAG}_ns_size_sweep.txt $O/${TAG}_ns_steps_sweep.txt; python3 -c "
import json; d=json.load(open('$O/${TAG}_bench_steps20.json')); print('step_us %.2f frac %.3f config5 %.3f' % (1000*d['ms_per_step'], d['roofline']['frac'], d['config5_single_gpu']['roofline_frac'])); print([ (s['workload'] if 'workload' in s else s.get('config',{}).get('workload','?'))[:30] + ' %.3f' % s['roofline']['frac'] for s in d['secondary']])"
