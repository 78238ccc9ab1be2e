#!/bin/bash
# round 3: launch chains x graph replay matrix on the default kernel (host enqueue vs GPU step)
set -u
O=gpurun_out/ns_chains; mkdir -p $O
for split in 1 2 3 4; do
  for g in "" "--graph"; do
    for steps in 20 200; do
      python3 bench.py --gpus 1 --steps $steps --warmup 5 --no-secondary --no-cpu-baseline --split $split $g > $O/s${split}_g${g#--}_k$steps.json 2>> $O/err.txt
      python3 - <<PY
import json
d=json.load(open("$O/s${split}_g${g#--}_k$steps.json"))
print("split $split graph '${g}' steps $steps: us/step %.2f  min %.2f p10 %.2f wall %.2f frac %.3f" % (d["ms_per_step"]*1e3, d["timing"]["ms_per_step_min"]*1e3, d["timing"]["ms_per_step_p10"]*1e3, d["timing"]["wall_ms_per_step_median"]*1e3, d["roofline"]["frac"]))
PY
    done
  done
done
