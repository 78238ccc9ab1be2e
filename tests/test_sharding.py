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
@pytest.mark.parametrize("workload,unit", [("aec", "frames/s"), ("bt1024", "macroblocks/s")])
def test_secondary_bench_under_torchrun_one_rank(workload, unit):
    """`bench.py --workload aec|bt1024 --gpus N` has the headline's shape (shard, barrier, MAX over ranks): rehearsed
    with one rank under torch.distributed.run."""
    import json
    import subprocess
    import sys

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--workload", workload, "--gpus", "1", "--steps", "40", "--warmup", "160", "--streams-per-gpu", "512",
           "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["unit"] == unit
    assert line["value"] > 0 and "x1" in line["config"]["parallelism"]


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


@pytest.mark.gpu
def test_two_shards_on_two_devices_aec_and_bt_vs_oracle():
    """The `device >= 1` path of AecBatch and BtBatch (their own tables, streams and state on the second GPU), and the
    caller's current device left alone by every entry point: two shards on two devices in one process against the
    oracles.  Skipped on a one-GPU box."""
    import torch

    from audiosignalprocess_amd.aec import AecBatch
    from audiosignalprocess_amd.bt import BtBatch
    from audiosignalprocess_amd.ns import device_count
    from audiosignalprocess_amd.synth import aec_frames, bt_samples
    from tests.oracle_lib import OracleAec, OracleBt

    if device_count() < 2:
        pytest.skip("one HIP device on this box (the 8-GPU curve is the driver's to measure)")
    torch.cuda.set_device(0)
    per, F = 3, 130
    outs = []
    for rank in (0, 1):
        s0, n = shard_streams(rank, 2, per)
        far, near = aec_frames(n, F, stream0=s0)
        g = AecBatch(n, device=rank)
        outs.append((g.run(far, near), far, near))
        g.close()
        assert torch.cuda.current_device() == 0      # AspDeviceScope: the caller's device is put back
    for out, far, near in outs:
        for s in range(per):
            want = OracleAec().run(far[:, s], near[:, s])
            num = np.sqrt(((out[:, s] - want).astype(np.float64) ** 2).sum())
            assert num <= 1e-5 * np.sqrt((want.astype(np.float64) ** 2).sum())
    for rank in (0, 1):
        s0, n = shard_streams(rank, 2, per)
        bt = BtBatch(n, 1024, device=rank)
        x = bt_samples(n, 3 * bt.macro, stream0=s0)
        y = bt.run(x)
        for ch in range(n):
            assert np.array_equal(y[ch].view(np.uint32), OracleBt(1024).run(x[ch]).view(np.uint32))
        assert torch.cuda.current_device() == 0
