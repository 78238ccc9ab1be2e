"""CPU suite: the N > 1 path.  Streams shard across ranks with no data-path collective; the only
communication is the timing barrier and a MAX over ranks.  Rehearsed with gloo, world_size 2:
each rank runs its shard through the oracle (the GPU library needs a device), and the union of
the shards must equal the single-process run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audiosignalprocess_amd.shard import max_over_ranks, shard_streams


def test_shard_streams_partition():
    for world in (1, 2, 4, 8):
        got = []
        for r in range(world):
            s0, n = shard_streams(r, world, 8192)
            assert n == 8192
            got.extend(range(s0, s0 + n))
        assert got == list(range(8192 * world))  # BASELINE config 5 at world = 8: 65 536 streams


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, per_rank, frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audiosignalprocess_amd.synth import ns_frames
    from tests.oracle_lib import OracleNs

    s0, n = shard_streams(rank, world, per_rank)
    x = ns_frames(n, frames, stream0=s0)
    y = OracleNs(n, policy=1).run(x)
    np.save(os.path.join(out_dir, "y%d.npy" % rank), y)
    t = max_over_ranks(dist, [float(rank + 1), 10.0 - rank])
    assert t == [float(world), 10.0]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shards_cover_the_batch(tmp_path):
    world, per_rank, frames = 2, 3, 40
    port = _free_port()
    mp.spawn(_worker, args=(world, port, per_rank, frames, str(tmp_path)), nprocs=world, join=True)
    from audiosignalprocess_amd.synth import ns_frames
    from tests.oracle_lib import OracleNs

    full = OracleNs(world * per_rank, policy=1).run(ns_frames(world * per_rank, frames))
    got = np.concatenate([np.load(tmp_path / ("y%d.npy" % r)) for r in range(world)], axis=1)
    assert np.array_equal(full, got)


# ---------------------------------------------------------------------------------------------
# GPU side of the N > 1 path: no multi-GPU node is available to these tests, so the
# code shape of `bench.py --gpus N` is rehearsed with ONE rank under torch.distributed.run (RCCL
# initialisation, NsBatch on the rank's own device, barrier, MAX over ranks), and -- where a second
# device exists -- two shards on two devices in one process against the oracle.
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank():
    import json
    import subprocess
    import sys

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "8", "--warmup", "3", "--streams-per-gpu", "2048", "--regions", "5",
           "--no-cpu-baseline", "--no-secondary"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["steps"] == 8
    assert line["config"]["streams_per_gpu"] == 2048 and line["config"]["total_streams"] == 2048
    assert line["value"] > 0 and line["timing"]["host_enqueue_us_per_step"] > 0


@pytest.mark.gpu
def test_two_shards_on_two_devices_vs_oracle():
    from audiosignalprocess_amd.ns import NsBatch, device_count
    from audiosignalprocess_amd.synth import ns_frames
    from tests.oracle_lib import REDUCE_TREE64P, OracleNs

    if device_count() < 2:
        pytest.skip("one HIP device on this box (the 8-GPU curve is the driver's to measure)")
    per, F = 12, 120
    batches, xs = [], []
    for rank in (0, 1):
        s0, n = shard_streams(rank, 2, per)
        xs.append(ns_frames(n, F, stream0=s0))
        batches.append(NsBatch(n, device=rank, policy=1))
    ys = [b.analyze_process(x) for b, x in zip(batches, xs)]
    full = OracleNs(2 * per, policy=1, reduce_mode=REDUCE_TREE64P).run(ns_frames(2 * per, F))
    assert np.array_equal(np.concatenate(ys, axis=1), full)
    for b in batches:
        b.close()
