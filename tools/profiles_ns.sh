#!/bin/bash
# Evidence set of the NS headline for profiles/ (one GPU session): the driver's bench line, rocprofv3 kernel
# stats of the same command, PMC instruction mix, fabric traffic (FETCH_SIZE / WRITE_SIZE in separate passes, at
# the bench size and at 32768 streams where every byte is HBM traffic: the calibration the guide asks for),
# phase stamps, launch timeline, batch-size sweep.   usage: tools/profiles_ns.sh <tag>   (writes gpurun_out/<tag>/)
export TMPDIR=/tmp
TAG=${1:-r03}; O=gpurun_out/$TAG; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_steps20.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.json 2> $O/stats.err
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/${TAG}_ns_kernel_stats.csv
ARGS="--steps 40 --warmup 260 --no-cpu-baseline --no-secondary --split 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 bench.py $ARGS > $O/p$i.json 2> $O/p$i.err || echo "pmc pass $i failed"
done
for S in 4096 32768; do
  R=100; [ $S = 32768 ] && R=12
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/t${S}_$C -- python3 bench.py --steps 30 --warmup 252 --no-cpu-baseline --no-secondary --split 1 --streams-per-gpu $S --ring $R > $O/t${S}_$C.json 2> $O/t${S}_$C.err || echo fail $S $C
  done
done
python3 - > $O/${TAG}_ns_pmc.txt <<PY
import csv,glob,collections
for d in sorted(glob.glob('$O/p*/')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'ns_frame1_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        w=sum(acc['SQ_WAVES'][-40:])/40 if acc.get('SQ_WAVES') else 4096.0
        for k,v in sorted(acc.items()):
            t=v[-40:]; print('%-28s per-launch %.4g  per-wave %.1f' % (k, sum(t)/len(t), sum(t)/len(t)/w))
PY
python3 - > $O/${TAG}_ns_traffic.txt <<PY
import csv,glob
print("fabric traffic of ns_frame1_kernel<false> per launch (one launch = one frame step of all streams, --split 1), KB per stream:")
res={}
for S in (4096,32768):
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob('$O/t%d_%s/*/*counter_collection.csv'%(S,C)):
            v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'ns_frame1_kernel' in r['Kernel_Name'] and r['Counter_Name']==C]
            t=v[-30:]; res[(S,C)]=sum(t)/len(t)/S; print(S,C,'per-launch %.6g'%(sum(t)/len(t)), 'per-stream %.4f'%(sum(t)/len(t)/S))
print(res)
PY
python3 tools/ns_stamps.py > $O/${TAG}_ns_stamps.txt 2>&1
python3 tools/ns_timeline.py 4096 2 40 > $O/${TAG}_ns_timeline.txt 2>&1
bash tools/ns_size_sweep.sh > $O/${TAG}_ns_size_sweep.txt 2>&1
SWEEP_S=8192 bash tools/ns_size_sweep.sh >> $O/${TAG}_ns_size_sweep.txt 2>&1
./tools/probe/bin/issue_probe3 > $O/${TAG}_issue_probe3.txt 2>&1
tail -3 $O/${TAG}_ns_traffic.txt; head -3 $O/${TAG}_ns_kernel_stats.csv | cut -c1-200; python3 -c "
import json; d=json.load(open('$O/${TAG}_bench_steps20.json')); print('step_us %.2f frac %.3f config5 %.3f' % (1000*d['ms_per_step'], d['roofline']['frac'], d['config5_single_gpu']['roofline_frac'])); print([ (s['workload'] if 'workload' in s else s.get('config',{}).get('workload','?'))[:30] + ' %.3f' % s['roofline']['frac'] for s in d['secondary']])"
