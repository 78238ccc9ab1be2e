#!/bin/bash
# LDS-side PMC pass of the AEC hand-off kernel, per wave (= per stream and WebRtcAec_Process call), and per CU-cycle
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_aec_lds}; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 bench.py --workload aec --steps 40 --warmup 160 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 bench.py --workload aec --steps 40 --warmup 160 --no-cpu-baseline > $OUT/p2.json 2> $OUT/p2.err
python3 - <<PY
import csv,glob,collections
for pas in ('p1','p2'):
    for f in glob.glob('$OUT/%s/*/*counter_collection.csv' % pas):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'aec_process_flow' in r['Kernel_Name']: acc[r['Counter_Name']].append((float(r['Counter_Value']), float(r['Grid_Size'])/64.0))
        for k,v in sorted(acc.items()):
            t=v[-3:]; print('  %-26s per launch %.4g   per wave %.1f' % (k, sum(x for x,_ in t)/len(t), sum(x/w for x,w in t)/len(t)))
PY

