"""CPU suite: the N > 1 path.  Streams shard across ranks with no data-path collective; the only
communication is the timing barrier and a MAX over ranks.  Rehearsed with gloo, world_size 2:
each rank runs its shard through the oracle (the GPU library needs a device), and the union of
the shards must equal the single-process run."""
import os
import socket

import numpy as np
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
