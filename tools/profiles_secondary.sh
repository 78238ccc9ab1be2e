#!/bin/bash
# round-2 evidence set for the secondary pipelines: bench lines, rocprofv3 kernel stats, PMC passes, phase stamps
export TMPDIR=/tmp
TAG=${1:-v2}
O=gpurun_out/profiles_$TAG; mkdir -p $O
for W in aec bt1024; do
  python3 bench.py --workload $W > $O/${ROUND:-r03}_${W}_${TAG}_bench.json 2> $O/${W}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${W}_stats -- python3 bench.py --workload $W --no-cpu-baseline > $O/${W}_stats.json 2> $O/${W}_stats.err
  f=$(ls $O/${W}_stats/*/*_kernel_stats.csv | head -1); cp $f $O/${ROUND:-r03}_${W}_${TAG}_kernel_stats.csv; head -4 $O/${ROUND:-r03}_${W}_${TAG}_kernel_stats.csv | cut -c1-200
done
./tools/pmc_aec.sh $O/pmc_aec > $O/${ROUND:-r03}_aec_${TAG}_pmc.txt 2>&1; tail -17 $O/${ROUND:-r03}_aec_${TAG}_pmc.txt | head -3
bash tools/traffic_sec.sh $O/traffic > $O/${ROUND:-r03}_traffic_sec.txt 2>&1; cat $O/${ROUND:-r03}_traffic_sec.txt
./tools/pmc_bt.sh ${ROUND:-r03}_$TAG > $O/${ROUND:-r03}_bt1024_${TAG}_pmc.txt 2>&1
python3 tools/bt_stamps.py > $O/${ROUND:-r03}_bt1024_${TAG}_stamps.txt 2>&1
python3 tools/aec_stamps.py > $O/${ROUND:-r03}_aec_${TAG}_stamps.txt 2>&1
python3 -c "
import json
for w in ('aec','bt1024'):
    d=json.load(open('$O/${ROUND:-r03}_%s_${TAG}_bench.json' % w)); print(w, 'step_us %.1f value %.3g frac %.3f cpu %s' % (1000*d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value')))
"
