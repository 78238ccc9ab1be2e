#!/usr/bin/env python3
"""bench.py -- throughput of the batched Wiener noise suppressor on MI355X.

Metric (BASELINE.json): 10 ms @16 kHz audio frames per second through the NS
hot path (Analyze + Process, policy 1), whole job over N GPUs, plus the
achieved fraction of the HBM roofline for the fused frame-step kernel.

One "step" = one 10 ms frame of every stream = ONE launch of the fused kernel
(frame-synchronous model: all per-stream state round-trips HBM every step).
Workload at N = 1: BASELINE config[1], 4096 concurrent mono 16 kHz streams on
one MI355X.  N > 1: the same 4096 streams on every GPU (weak scaling; streams
are independent, no collective on the data path -- torch.distributed/RCCL is
used only for the timing barrier and the max-over-ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_FRAME = 15716  # frame-synchronous NS model
# HBM bytes per stream-frame of ns_frame_kernel<true,true> from the PMC counters FETCH_SIZE /
# WRITE_SIZE (separate rocprofv3 --pmc passes, tools/traffic_ns.sh), calibrated on the kernel's own
# known byte count at 32768 streams as MI355X_MICROARCH.md prescribes for access widths it does not
# cover; see profiles/README.md ("HBM traffic of the NS kernel").  Measured once per round, not live.

# Secondary lines: fabric-side bytes from the same counters (tools/traffic_sec.sh, profiles/r04_traffic_sec.txt),
# stored per unit of work, not measured in the run (hand-off builds, as benched).  BT-1024: 2 x 10.060 KB FETCH_SIZE +
# 18.075 KB WRITE_SIZE per macroblock (8-byte lanes: the guide's factor 2 on reads; the input tail is written once per
# launch, not per macroblock: 0.93 x the algorithmic 40 960 B); BT-256: 2 x 2.395 KB + 4.519 KB per macroblock.  AEC:
# 26.01 KB FETCH_SIZE and 37.63 KB WRITE_SIZE per stream and frame (the frame's far-end work included) of
# 4-byte-per-lane accesses (a width the guide does not calibrate; with its factor 2 on reads 1.29 x the algorithmic
# 71 400 B: FilterAdaptation reads the far history and the filter a second time in a call's first block, from the
# Infinity Cache).
BT_TRAFFIC_BYTES_PER_MACROBLOCK = (2 * 10.0597 + 18.0747) * 1024
BT256_TRAFFIC_BYTES_PER_MACROBLOCK = (2 * 2.3950 + 4.5187) * 1024
AEC_TRAFFIC_BYTES_PER_FRAME = (2 * 26.0074 + 37.6261) * 1024

# ns_frame1_kernel<false> (the benched kernel, round 3: profiles/r03_ns_traffic.txt): FETCH_SIZE 3.8648 /
# WRITE_SIZE 7.2188 KB per stream and launch at 4096 streams against 3.8233 / 7.2188 KB at 32768 streams, where
# the kernel's known traffic is all HBM traffic: 7 808 B read (12 rows of 512 B, two 384-B sliding buffers, 256 B
# of scalars, 640 B of samples) -- the calibration of its 8-byte-per-lane reads -- and 7 392 B written (the
# quantile row is written back only when it changes: 7 296 B, + 96 B of histogram atomics), which WRITE_SIZE
# reports exactly.  The one-stream kernel of the two-call protocol keeps its round-1 constant.
PMC_TRAFFIC_BYTES_PER_FRAME = (4.832 * 1.638 + 9.158 / 1.09) * 1024      # ns_frame_kernel<true,true>
PMC_TRAFFIC_BYTES_PER_FRAME_PAIR = 7808 * (3.8648 / 3.8233) + 7.21875 * 1024
# the hand-off build ns_frame1_kernel<false, true> (profiles/r04_ns_traffic.txt): FETCH_SIZE 3.6010 KB per stream-frame at
# 4096 streams against 3.8292 KB at 32768 streams, where every one of the kernel's 7 808 read bytes is HBM traffic;
# WRITE_SIZE 7.2567 KB (7 392 B of state and samples, the step counter, partial lines)
PMC_TRAFFIC_BYTES_PER_FRAME_FLOW = 7808 * (3.6010 / 3.8292) + 7.2567 * 1024
NS_PRIME_FRAMES = 250  # untimed set-up frames + warm-up >= this (start-up phase of ns_core.c is 200)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return int(os.environ.get("ASP_BENCH_CPU_THREADS", n))


def cpu_baseline(seconds_budget=20.0):
    """The CPU path timed on this box's host cores (rank 0, N = 1 only).

    kind "reference": oracle/_ref/libns_ref.so = the reference's own ns_core.c /
    fft4g.c compiled in the build container; kind "port": this repo's C
    restatement (oracle/ns_oracle.c) when the reference build is absent.
    """
    from audiosignalprocess_amd.synth import ns_frames
    from tests import oracle_lib

    threads = host_cores()
    warm, timed = 250, 2000   # 1024 streams x 2000 frames: about 15 core-seconds of the reference's code
    # ~90 k frames/s/core measured for the reference
    streams = int(max(threads, min(64 * threads, seconds_budget * 90e3 / (warm + timed))))
    x = ns_frames(streams, 50, frame0=50)
    reps_w, reps_t = warm // 50, timed // 50
    if oracle_lib.have_ref():
        eng, kind = oracle_lib.RefNs(streams, policy=1), "reference"
    else:
        eng, kind = oracle_lib.OracleNs(streams, policy=1), "port"
    for _ in range(reps_w):
        eng.run(x, threads=threads)
    t0 = time.perf_counter()
    for _ in range(reps_t):
        eng.run(x, threads=threads)
    dt = time.perf_counter() - t0
    # SURVEY 8(d) asks for both: the same code on ONE thread (a bounded sample of its own: 16 streams x 500 frames,
    # about 0.1 core-seconds past its own warm-up)
    s1, timed1 = 16, 500
    x1 = np.ascontiguousarray(x[:, :s1])
    eng1 = oracle_lib.RefNs(s1, policy=1) if kind == "reference" else oracle_lib.OracleNs(s1, policy=1)
    for _ in range(reps_w):
        eng1.run(x1, threads=1)
    t1 = time.perf_counter()
    for _ in range(timed1 // 50):
        eng1.run(x1, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": streams * timed / dt,
        "unit": "frames/s",
        "cores": threads,
        "kind": kind,
        "sample": "%d streams x %d frames after %d warm-up frames, %d pthreads over streams, "
                  "gcc -O2 -ffp-contract=off" % (streams, timed, warm, threads),
        "single_thread": {"value": s1 * timed1 / dt1, "unit": "frames/s", "cores": 1,
                          "sample": "%d streams x %d frames after %d warm-up frames, one thread" % (s1, timed1, warm)},
    }


def init_ranks(args):
    """(rank, world, local_rank, dist) of this process; with more than one rank (or under torch.distributed.run)
    the RCCL process group is initialised: it serves the timing barrier and the MAX over ranks only."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher" % args.gpus)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:
        # launched by torch.distributed.run: the same code shape at every N (a one-rank launch rehearses
        # the RCCL initialisation, the barrier and the MAX over ranks of the N-rank run)
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return rank, world, local_rank, dist


def rank_barrier(dist):
    import torch

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def finish_ranks(dist):
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_bt(args):
    """Secondary line (BASELINE config 3 / 1): BlockThresholding macroblocks per second, 1 GPU."""
    import torch

    from audiosignalprocess_amd.bt import BtBatch
    from audiosignalprocess_amd.synth import bt_samples

    from audiosignalprocess_amd.shard import max_over_ranks, shard_streams

    rank, world, local_rank, dist = init_ranks(args)
    n = 1024 if args.workload == "bt1024" else 256
    # every rank owns a disjoint contiguous shard of stream-channels on its own GPU (no data-path collective)
    stream0, S = shard_streams(rank, world, args.streams_per_gpu)
    ring = 4
    g = BtBatch(S, n, device=local_rank)
    x = bt_samples(S, ring * g.macro, stream0=stream0).reshape(S, ring, g.macro).transpose(1, 0, 2)
    d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d_out = torch.empty_like(d_in)
    # timed region: `--steps` macroblock steps (the default 1000 = 117 ms at N = 1024).  A region costs 1-2 ms however long it
    # is (the chip's clock ramps up after the idle gap of the barrier; both builds show it: profiles/r04_region_length.txt),
    # so the 13 ms regions of earlier rounds (--steps / 10) read 12 % slower than the sustained rate
    steps, warm = max(args.steps, 10), max(args.warmup // 10, 4)
    g.timed_steps(d_in.data_ptr(), d_out.data_ptr(), ring, warm)
    rank_barrier(dist)
    t0 = time.perf_counter()
    ev_ms = g.timed_steps(d_in.data_ptr(), d_out.data_ptr(), ring, steps)
    rank_barrier(dist)
    wall = time.perf_counter() - t0
    ev_ms, wall = max_over_ranks(dist, [ev_ms, wall], device="cuda")   # the slowest rank's region
    if rank != 0:
        return finish_ranks(dist)
    bt_flow = os.environ.get("ASP_BT_FLOW", "1")[:1] != "0" and (n == 1024 or S % 4 == 0) and steps >= 2   # bt_api.hip, bt_flow_applies
    algo = 40 * (n // 2) * 2 * 1  # 10 B per sample: in + out + both tails read and written
    algo = 10 * g.macro
    launch_s = ev_ms / 1e3 / steps             # one clock (hipEvents on the launch stream) for every number of the line
    achieved = algo * S / launch_s / 1e9
    line = {
        "metric": "BlockThresholding macroblocks/sec (secondary)", "value": S * world / launch_s,
        "unit": "macroblocks/s", "n_gpus": world, "steps": steps, "warmup": warm,
        "ms_per_step": 1e3 * launch_s, "wall_ms_per_step": 1e3 * wall / steps,
        "parity": "unpinned (the reference's kiss_fft does not compile: _kiss_fft_guts.h is absent; DESIGN.md section 2)",
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Denoise/BlockThresholding: %d-pt STFT, Stein block threshold, %d "
                               "stream-channels per MI355X (%d on %d GPU%s), one 8-hop macroblock per stream-channel and step"
                               % (n, S, S * world, world, "" if world == 1 else "s"),
                   "samples_per_s": S * world * g.macro / launch_s,
                   "parallelism": "stream-sharded x%d, no collectives" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": (BT_TRAFFIC_BYTES_PER_MACROBLOCK if n == 1024 else BT256_TRAFFIC_BYTES_PER_MACROBLOCK) * S,
                     "traffic_source": "stored constant: PMC passes kept under profiles/r04_traffic_sec.txt (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), per step of all stream-channels; not measured in this run",
                     "kernel": ("bt_macroblock8_kernel<false, %s>" if n == 1024 else "bt_macroblock8_kernel<true, %s> (four stream-channels per workgroup)") % ("true" if bt_flow else "false"),
                     "launch_chains": 1 if bt_flow else 2 if S >= 2048 else 1,
                     "macroblock_steps_per_launch": min(steps, 64) if bt_flow else 1,
                     "algorithmic_bytes_per_step": algo * S, "avg_step_us": launch_s * 1e6},
    }
    if not args.no_cpu_baseline and world == 1:
        # the CPU restatement (oracle/bt_oracle.c; the reference itself is unbuildable here, DESIGN.md
        # section 2), one stream-channel per thread; ctypes releases the GIL inside the C calls
        from concurrent.futures import ThreadPoolExecutor

        from tests.oracle_lib import OracleBt
        cores = host_cores()
        per, reps = (100, 30) if n == 1024 else (400, 30)
        blocks = per * reps
        xs = bt_samples(cores, per * g.macro)

        def one(ch):
            o = OracleBt(n)
            for _ in range(reps):
                o.run(xs[ch])
        OracleBt(n).run(xs[0][:g.macro])          # table construction outside the timed region
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, range(cores)))
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": cores * blocks / dt, "unit": "macroblocks/s", "cores": cores,
                                "kind": "port",
                                "sample": "%d stream-channels x %d macroblocks through oracle/bt_oracle.c, "
                                          "one thread each" % (cores, blocks)}
    print(json.dumps(line), flush=True)
    finish_ranks(dist)


def bench_split(args):
    """Secondary line: the three-band SplittingFilter of a 48 kHz channel, 10 ms frames
    through Analysis (48 -> 64 kHz sinc resampler + two QMF stages) and Synthesis (the inverse), 1 GPU."""
    import torch

    from audiosignalprocess_amd.qmf import MEM_DEVICE, SplitBatch, _check

    Cn = args.streams_per_gpu
    ring = 8
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((ring, Cn, 480)) * 3000).astype(np.int16)
    d_in = torch.from_numpy(x).cuda()
    d_bands = torch.empty((3, Cn, 160), dtype=torch.int16, device="cuda")
    d_out = torch.empty_like(d_in)
    g = SplitBatch(Cn, 3)

    def step(k):
        _check(g.lib.AspSplitBatch_Analysis(g.h, d_in[k % ring].data_ptr(), d_bands.data_ptr(), MEM_DEVICE), "Analysis")
        _check(g.lib.AspSplitBatch_Synthesis(g.h, d_bands.data_ptr(), d_out[k % ring].data_ptr(), MEM_DEVICE), "Synthesis")
    steps, warm = max(args.steps // 4, 10), max(args.warmup // 10, 10)
    for k in range(warm):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    algo = 2 * (480 * 2 + 3 * 160 * 2)      # samples in / bands out and back, int16; filter state is a few hundred bytes more
    achieved = algo * Cn * steps / wall / 1e9
    line = {
        "metric": "48 kHz three-band split + merge, 10 ms channel-frames/sec (secondary)", "value": Cn * steps / wall,
        "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * wall / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "s16", "data": "synthetic",
        "config": {"workload": "SplittingFilter (splitting_filter.cc:28-170): %d channels at 48 kHz, Analysis + "
                               "Synthesis per 10 ms frame, 1 MI355X" % Cn},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "sinc_resample_kernel + qmf_analysis4_kernel / qmf_synthesis4_kernel (six launches per step; "
                               "wall-clock figure, launch-bound)",
                     "algorithmic_bytes_per_launch": algo * Cn, "avg_launch_us": None},
    }
    print(json.dumps(line), flush=True)


def aec_flow_active(ext, dmode, steps):
    """Does AspAecBatch_TimedSteps run the hand-off build?  (aec_api.hip, aec_flow_applies)"""
    fused = os.environ.get("ASP_AEC_AGN_FUSED", "1")[:1] != "0"   # aec_api.hip, agn_fused
    return os.environ.get("ASP_AEC_FLOW", "1")[:1] != "0" and (dmode in ("off", "logging") or fused) and steps >= 2


def bench_aec(args):
    """Secondary line (BASELINE config 4): 10 ms / 16 kHz frames per second through the echo
    canceller (WebRtcAec_BufferFarend + WebRtcAec_Process per frame), 1 GPU."""
    import torch

    from audiosignalprocess_amd.aec import AecBatch
    from audiosignalprocess_amd.synth import aec_frames

    from audiosignalprocess_amd.shard import max_over_ranks, shard_streams

    rank, world, local_rank, dist = init_ranks(args)
    # every rank owns a disjoint contiguous shard of streams on its own GPU (no data-path collective; the host
    # control plane is per batch and simply replicated per rank)
    stream0, S = shard_streams(rank, world, args.streams_per_gpu)
    ring = 40
    far1, near1 = aec_frames(64, ring, stream0=(stream0 % 64))   # 64 distinct streams, tiled over the shard
    idx = np.arange(S) % 64
    d_far = torch.from_numpy(np.ascontiguousarray(far1[:, idx])).cuda()
    d_near = torch.from_numpy(np.ascontiguousarray(near1[:, idx])).cuda()
    d_out = torch.empty_like(d_near)
    g = AecBatch(S, 16000, device=local_rank)
    ext = bool(getattr(args, "aec_extended", False))
    if ext:     # WebRtcAec_enable_delay_correction: the 32-partition extended filter
        g.enable_delay_correction(1)
    dmode = getattr(args, "aec_delay", "off")
    if dmode != "off":      # delay logging (hand-off build: one estimator launch per process launch; else one per frame) / the delay-agnostic mode (per sub-frame)
        assert g.set_config(1, delay_logging=1) == 0
        if dmode == "agnostic":
            g.enable_reported_delay(0)
    # warm-up passes the start-up phase; timed region: `--steps` frames (the default 1000 = 79 ms; the 20 ms regions of
    # earlier rounds, --steps / 4, carry the same ~1 ms per-region cost as the BT line: profiles/r04_region_length.txt)
    steps, warm = max(args.steps, 10), max(args.warmup // 2, 80)
    g.timed_steps(d_far.data_ptr(), d_near.data_ptr(), d_out.data_ptr(), 160, ring, warm)
    rank_barrier(dist)
    assert g.control().startup_phase == 0
    t0 = time.perf_counter()
    ev_ms = g.timed_steps(d_far.data_ptr(), d_near.data_ptr(), d_out.data_ptr(), 160, ring, steps)
    rank_barrier(dist)
    wall = time.perf_counter() - t0
    assert bool(torch.isfinite(d_out).all())
    ev_ms, wall = max_over_ranks(dist, [ev_ms, wall], device="cuda")   # the slowest rank's region
    if rank != 0:
        return finish_ranks(dist)
    flow = aec_flow_active(ext, dmode, steps)
    # 2.5 blocks x 28 560 B per 10 ms frame; the same accounting with 32 partitions:
    # read X, W 2 x 32 x 130 + 1 229 floats, write W 32 x 130 + 1 231 floats = 59 760 B per block
    algo = 149400 if ext else 71400
    step_s = ev_ms / 1e3 / steps
    achieved = algo * S / step_s / 1e9
    line = {
        "metric": "AEC 10 ms frames/sec (secondary)", "value": S * world / step_s, "unit": "frames/s",
        "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * step_s,
        "wall_ms_per_step": 1e3 * wall / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "WebRTC AEC (test_aec_module): 10 ms/16 kHz far+near frames, %d concurrent "
                               "streams per MI355X (%d on %d GPU%s), %d partitions, the far-end work fused into the "
                               "process launch of each frame%s%s"
                               % (S, S * world, world, "" if world == 1 else "s", 32 if ext else 12,
                                  "" if dmode == "off" else "; delay estimation: " + dmode,
                                  "; hand-off build: %d frame steps per launch" % min(steps, 64) if flow else ""),
                   "parallelism": "stream-sharded x%d, no collectives" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None if (ext or dmode != "off") else AEC_TRAFFIC_BYTES_PER_FRAME * S,
                     "traffic_source": "stored constant: PMC passes kept under profiles/r04_traffic_sec.txt (2 x FETCH_SIZE + WRITE_SIZE; 4-byte-per-lane accesses, a width MI355X_MICROARCH.md does not calibrate), per frame step of all streams; not measured in this run",
                     "kernel": ("aec_process_flow_kernel" if flow else
                                "aec_process_agn_kernel" if dmode == "agnostic" and os.environ.get("ASP_AEC_AGN_FUSED", "1")[:1] != "0"
                                else "aec_process_kernel") + " (the far-end work of the frame inside it"
                               + (", the stream's far-buffer control steps and delay estimator too" if dmode == "agnostic" and (flow or os.environ.get("ASP_AEC_AGN_FUSED", "1")[:1] != "0") else "")
                               + ("; aec_delay_bits_kernel once per launch" if flow and dmode == "logging" else "") + ")",
                     "launch_chains": 1 if flow else 2 if S >= 2048 and dmode == "off" else 1,
                     "frame_steps_per_launch": min(steps, 64) if flow else 1,
                     "algorithmic_bytes_per_step": algo * S, "avg_step_us": step_s * 1e6},
    }
    if not args.no_cpu_baseline and world == 1:
        from tests import oracle_lib
        cores = host_cores()
        if oracle_lib.have_aec_ref():
            # the reference's own aec_core.c / aec_rdft.c / echo_cancellation.c compiled in the build container
            # (oracle/_ref/libaec_ref.so, plain-C path): one handle per thread, ctypes releases the GIL
            from concurrent.futures import ThreadPoolExecutor
            per, Fc = 8, 1500   # 128 streams x 1500 frames: about 10 core-seconds
            Sc = per * cores
            farc, nearc = aec_frames(Sc, Fc)
            engs = [[oracle_lib.RefAec(16000) for _ in range(per)] for _ in range(cores)]   # created one by one

            def one(t):
                for i, e in enumerate(engs[t]):
                    s_ = t * per + i
                    e.run(np.ascontiguousarray(farc[:, s_]), np.ascontiguousarray(nearc[:, s_]))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(one, range(cores)))
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": Sc * Fc / dt, "unit": "frames/s", "cores": cores, "kind": "reference",
                                    "sample": "%d streams x %d frames through the compiled reference (oracle/_ref/"
                                              "libaec_ref.so, gcc -O2 -ffp-contract=off, plain-C path), %d threads"
                                              % (Sc, Fc, cores)}
        else:
            Sc, Fc = 16 * cores, 300
            farc, nearc = aec_frames(Sc, Fc)
            t0 = time.perf_counter()
            oracle_lib.aec_oracle_run_mt(farc, nearc, threads=cores)
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": Sc * Fc / dt, "unit": "frames/s", "cores": cores, "kind": "port",
                                    "sample": "%d streams x %d frames through oracle/aec_oracle.c (bit-exact "
                                              "restatement of the reference), %d pthreads" % (Sc, Fc, cores)}
    print(json.dumps(line), flush=True)
    finish_ranks(dist)


def copy_ceiling_gbs(device=0):
    """The box's measured streaming-copy rate: a hand-written float4 copy of 1 GiB (read + written
    bytes, hipEvent-timed, far beyond the 256 MiB Infinity Cache) and of 32 MiB (the size of one
    NS frame step's traffic, which the Infinity Cache holds)."""
    from audiosignalprocess_amd.ns import copy_ceiling_gbs as cc

    return cc(1 << 30, 8, device), cc(32 << 20, 64, device)


def _secondary(args, workload, cpu_baseline=True):
    """One secondary workload (AEC / BT-1024) in a child process of the same run, so its allocations
    and library state never touch the headline measurement; returns its JSON line as a dict."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(args.secondary_steps),
           "--warmup", str(args.secondary_warmup)]
    if args.no_cpu_baseline or not cpu_baseline:
        cmd.append("--no-cpu-baseline")
    try:
        # a plain single-process child, also when this process was started by torch.distributed.run
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE")}
        out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600, env=env).stdout.strip().splitlines()
        return json.loads(out[-1])
    except Exception as e:  # noqa: BLE001  (a failed secondary must not take the headline down)
        return {"workload": workload, "error": str(e)[:300]}


FLOW_MAX_STEPS = 64  # kFlowMaxSteps of ns_api.hip: frame steps per launch of the hand-off build
FLOW_KERNELS = (2, 3)   # kernels with a hand-off build


def flow_active(args):
    """Does the K-step timed region run the hand-off build?  (ns_api.hip, flow_applies)"""
    if args.flow == "off" or (args.flow == "auto" and os.environ.get("ASP_NS_FLOW", "1")[:1] == "0"):
        return False
    return args.steps >= 2 and (args.kernel or 3) in FLOW_KERNELS and not args.graph


def ns_measure(args, S, rank, world, local_rank, dist):
    """Prime + warm up + time the fused NS frame step of S streams on this rank.  Returns the
    per-region hipEvent and wall times (seconds, length R) of `args.steps` steps each."""
    import torch

    from audiosignalprocess_amd.ns import NsBatch
    from audiosignalprocess_amd.shard import shard_streams
    from audiosignalprocess_amd.synth import ns_frames

    ring = args.ring
    # every rank owns a disjoint contiguous shard of stream ids (no data-path collective)
    stream0, S = shard_streams(rank, world, S)
    x = ns_frames(S, ring, stream0=stream0, frame0=50)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    del x
    ns = NsBatch(S, device=local_rank, policy=1, kernel=args.kernel or None)
    ns.set_graph(args.graph)
    if args.split > 1:
        ns.set_split(args.split)
    ns.set_flow({"auto": -1, "off": 0, "on": 1}[args.flow])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up: every stream is primed past the suppressor's start-up (END_STARTUP_LONG = 200 frames,
    # ns/defines.h:20: different, heavier branches) whatever --warmup is, so the timed steps are
    # the steady state of a long-running stream; the W warm-up steps then follow as asked
    primed = max(0, NS_PRIME_FRAMES - args.warmup)
    done = 0
    chunk = min(ring, max(args.steps, 2))   # launches of the timed region's shape (hand-off build: K steps per launch)
    while done < primed:
        n = min(chunk, primed - done)
        ns.analyze_process_device(d_in.data_ptr(), d_out.data_ptr(), n)
        done += n
    barrier()
    if args.warmup > 0:
        ns.timed_steps(d_in.data_ptr(), d_out.data_ptr(), ring, args.warmup)
    # the timed region: EXACTLY --steps fused frame steps between two barriers (+ device
    # synchronisation), repeated R times so the median is stable at small --steps; each region is
    # timed by hipEvents recorded on the launch stream around the K steps (the clock every number of
    # the line uses) and by the host's wall clock around the same region (reported beside it)
    R = args.regions if args.regions > 0 else max(5, min(200, -(-4000 // max(args.steps, 1))))
    ev, wall, enq = [], [], []
    ns.timed_steps(d_in.data_ptr(), d_out.data_ptr(), ring, args.steps)   # graph capture of the K-step region
    for _ in range(R):
        barrier()
        t0 = time.perf_counter()
        ev_ms = ns.timed_steps(d_in.data_ptr(), d_out.data_ptr(), ring, args.steps)
        barrier()
        wall.append(time.perf_counter() - t0)
        ev.append(ev_ms / 1e3)
        enq.append(ns.last_enqueue_us() / max(args.steps, 1))
    if not torch.isfinite(d_out).all():
        raise SystemExit("non-finite output")
    ns.close()
    del d_in, d_out
    ns_measure.host_enqueue_us_per_step = float(np.median(enq))  # this rank's host budget per step
    return np.array(ev), np.array(wall), S, primed


def pcie_inclusive(S, device):
    """The same fused step fed from and returning to pageable host buffers (ASP_MEM_HOST: copy in, K frame steps,
    copy out per call): what a caller that keeps its audio on the host sees.  Never `value`."""
    from audiosignalprocess_amd.ns import NsBatch
    from audiosignalprocess_amd.synth import ns_frames
    F, reps = 50, 4
    try:
        x = ns_frames(S, F, frame0=50)
        g = NsBatch(S, device=device, policy=1)
        g.set_split(2)
        for _ in range(2):
            g.analyze_process(x)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.analyze_process(x)
        dt = time.perf_counter() - t0
        g.close()
        return {"value": S * F * reps / dt, "unit": "frames/s", "streams": S,
                "GB_per_s_each_way": S * F * reps * 640 / dt / 1e9,
                "sample": "%d calls of %d frames x %d streams, pageable host buffers in and out, wall clock" % (reps, F, S)}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "error": str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=250)
    ap.add_argument("--streams-per-gpu", type=int, default=0,
                    help="default: 4096 at --gpus 1 (BASELINE config 2), 8192 at --gpus > 1 (config 5: 65 536 streams on 8 GPUs)")
    ap.add_argument("--ring", type=int, default=100, help="distinct input frames resident in HBM")
    ap.add_argument("--regions", type=int, default=0, help="repeats of the K-step timed region (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the AEC / BT-1024 / config-5 lines of the N = 1 run")
    ap.add_argument("--graph", action="store_true", help="replay captured hipGraphs (one linear graph per chain) instead of plain launches")
    ap.add_argument("--no-graph", action="store_true", help="(default) plain kernel launches")
    ap.add_argument("--secondary-steps", type=int, default=1000)
    ap.add_argument("--secondary-warmup", type=int, default=250)
    ap.add_argument("--split", type=int, default=2, help="sub-launches per fused step (1..4) of the plain build (--flow off)")
    ap.add_argument("--flow", default="auto", choices=["auto", "off", "on"],
                    help="hand-off build of the K-step region (include/asp_ns.h, AspNsBatch_SetFlow): up to 64 frame steps per "
                         "launch, a per-stream step counter in memory orders a stream's consecutive steps; auto = the library's "
                         "default (on; ASP_NS_FLOW=0 turns it off)")
    ap.add_argument("--kernel", type=int, default=0, choices=[0, 1, 2, 3],
                    help="fused-step kernel: 0 / 3 = one stream per wave, pair layout (ns_kernels1.hip, the default), "
                         "1 = one stream per wave, bins q / q + 64 (ns_kernels.hip)")
    ap.add_argument("--aec-extended", action="store_true", help="--workload aec: the 32-partition extended filter")
    ap.add_argument("--aec-delay", default="off", choices=["off", "logging", "agnostic"],
                    help="--workload aec: delay logging, or the delay-agnostic mode (per-stream far-buffer control)")
    ap.add_argument("--workload", default="ns", choices=["ns", "bt1024", "bt256", "aec", "split48"],
                    help="ns = the headline metric (default); bt* / aec / split48 = secondary lines")
    args = ap.parse_args()
    if args.streams_per_gpu <= 0:
        args.streams_per_gpu = 4096 if (args.gpus == 1 or args.workload != "ns") else 8192
    if args.workload == "aec":
        return bench_aec(args)
    if args.workload == "split48":
        return bench_split(args)
    if args.workload != "ns":
        return bench_bt(args)

    rank, world, local_rank, dist = init_ranks(args)
    import torch

    from audiosignalprocess_amd.shard import max_over_ranks

    ev, wall, S, primed = ns_measure(args, args.streams_per_gpu, rank, world, local_rank, dist)
    # per region: the slowest rank's time (MAX over ranks), then the median over regions
    ev = np.array(max_over_ranks(dist, list(ev), device="cuda"))
    wall = np.array(max_over_ranks(dist, list(wall), device="cuda"))

    if rank == 0:
        K = max(args.steps, 1)
        step_s = float(np.median(ev)) / K            # THE clock of this line: hipEvents over the K-step region
        frames_per_region = S * world * args.steps
        achieved = ALGO_BYTES_PER_FRAME * S / step_s / 1e9
        flow = flow_active(args)
        steps_per_launch = min(max(args.steps, 1), FLOW_MAX_STEPS) if flow else 1
        conc = 1 if flow else args.split
        kernel = {1: "ns_frame_kernel<true,true>", 2: "ns_frame2_kernel<false, true>" if flow else "ns_frame2_kernel<false, false>",
                  3: "ns_frame1_kernel<false, true>" if flow else "ns_frame1_kernel<false, false>"}[args.kernel or 3]
        line = {
            "metric": "audio frames/sec (10 ms @16 kHz) Wiener NS",
            "value": frames_per_region / (step_s * K),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * step_s,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "WebRTC NS (test_ns_module): 10 ms/16 kHz frames, %d concurrent mono "
                            "streams per MI355X (%d in total on %d GPU%s), policy 1, Analyze+Process fused, "
                            "frame-synchronous (every step reads and writes the whole state through memory; %s)"
                            % (S, S * world, world, "" if world == 1 else "s",
                               "hand-off build: %d frame steps per launch, a stream's consecutive steps ordered by a "
                               "step counter in memory" % steps_per_launch if flow else "1 launch per frame and sub-launch"),
                "streams_per_gpu": S,
                "total_streams": S * world,
                "input_ring_frames": args.ring,
                "primed_frames_in_setup": primed,
                "sub_launches_per_step": conc,
                "frame_steps_per_launch": steps_per_launch,
                "launch": ("hipGraph replay (one linear graph per chain, kernel nodes only)" if args.graph else "plain launches"),
                "parallelism": "stream-sharded x%d, no collectives" % world,
            },
            # one clock for value, ms_per_step and roofline: the median over `regions` repeats of the
            # K-step region of the hipEvent time recorded on the launch stream (max over ranks per region);
            # the host wall clock around the same regions is printed beside it
            "timing": {
                "clock": "hipEvent on the launch stream, median over regions (max over ranks per region)",
                "regions": int(len(ev)),
                "ms_per_step_min": 1e3 * float(ev.min()) / K,
                "ms_per_step_p10": 1e3 * float(np.percentile(ev, 10)) / K,
                "ms_per_step_p90": 1e3 * float(np.percentile(ev, 90)) / K,
                "ms_per_step_max": 1e3 * float(ev.max()) / K,
                "wall_ms_per_step_median": 1e3 * float(np.median(wall)) / K,
                "host_enqueue_us_per_step": ns_measure.host_enqueue_us_per_step,
                "wall_value_median": frames_per_region / float(np.median(wall)),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (PMC_TRAFFIC_BYTES_PER_FRAME_FLOW if flow else PMC_TRAFFIC_BYTES_PER_FRAME_PAIR if (args.kernel or 3) == 3
                            else PMC_TRAFFIC_BYTES_PER_FRAME) * S * steps_per_launch,
                "traffic_source": "stored constant for the kernel named in this line: PMC passes kept under "
                                  "profiles/r04_ns_traffic.txt (FETCH_SIZE / WRITE_SIZE in separate passes, reads "
                                  "calibrated at 32768 streams; not measured in this run), per launch",
                "kernel": kernel,
                # one frame step = `concurrent_launches` launches of this kernel side by side (one per HIP
                # stream, S / concurrent_launches streams each); achieved = algorithmic bytes of the step /
                # ms_per_step of this line
                "concurrent_launches": conc,
                "frame_steps_per_launch": steps_per_launch,
                "algorithmic_bytes_per_step": ALGO_BYTES_PER_FRAME * S,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * S * steps_per_launch // max(conc, 1),
                "avg_launch_us": step_s * 1e6 * steps_per_launch,
            },
        }
        if world == 1:
            try:
                cc, cc_small = copy_ceiling_gbs(local_rank)
                line["roofline"]["copy_ceiling"] = cc
                line["roofline"]["copy_ceiling_32MiB_in_infinity_cache"] = cc_small
                line["roofline"]["frac_of_copy_ceiling"] = achieved / cc
            except Exception as e:  # noqa: BLE001
                line["roofline"]["copy_ceiling"] = None
                line["roofline"]["copy_ceiling_error"] = str(e)[:200]
        if world == 1 and not args.no_secondary:
            # BASELINE config 5 on one GPU (8192 streams): the like-for-like N = 1 point of the 8-GPU run
            a2 = argparse.Namespace(**vars(args))
            a2.steps, a2.warmup, a2.regions = min(args.steps, 200), min(args.warmup, 50), 0
            ev5, wall5, S5, _ = ns_measure(a2, 8192, 0, 1, local_rank, None)
            s5 = float(np.median(ev5)) / max(a2.steps, 1)
            line["config5_single_gpu"] = {
                "workload": "8192 streams on 1 MI355X (the per-GPU share of BASELINE config 5)",
                "value": S5 / s5, "unit": "frames/s", "ms_per_step": 1e3 * s5, "steps": a2.steps, "regions": int(len(ev5)),
                "roofline_frac": ALGO_BYTES_PER_FRAME * S5 / s5 / 1e9 / HBM_PEAK_GBS,
                # what one rank of the 8-GPU run needs from its host core per step, beside the device time
                "host_enqueue_us_per_step": ns_measure.host_enqueue_us_per_step,
            }
        if world == 1 and not args.no_secondary:
            line["pcie_inclusive"] = pcie_inclusive(args.streams_per_gpu, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.no_secondary:
            torch.cuda.empty_cache()
            line["secondary"] = [_secondary(args, "aec"), _secondary(args, "bt1024"),
                                 _secondary(args, "bt256", cpu_baseline=False)]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
