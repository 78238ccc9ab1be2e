#!/bin/bash
# the driver's multi-process launch shape with one rank (rendezvous, nccl init, barrier, max over ranks)
export TMPDIR=/tmp
O=gpurun_out/r02_torchrun1; mkdir -p $O
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/out.json 2> $O/err.txt; echo rc=$?
tail -c 600 $O/out.json; tail -3 $O/err.txt
