#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X IQ->packets receive path.

Workload (BASELINE.json configs[1], SURVEY.md §8d cfg #2): single-channel BTLE GFSK demod +
access-address correlate + de-whiten/CRC decode on 1e9 synthetic cf32 IQ samples per GPU, input
already resident in HBM when the timed region starts.  A "step" is one pass of the whole path
over that batch, packet records landed in host memory.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: capture segments shard across ranks with no data-path collective (weak scaling: every
rank processes its own 1e9-sample segment); decoded packet records are gathered to rank 0 with
RCCL inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the stream the
dominant kernel runs on; `cpu_baseline` times the CPU oracle (oracle/, kind "port") on a bounded
sample of the same workload on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TILE = 1 << 22                  # samples in the host-generated, noise-free packet tile
SIGMA = 0.05


def make_workload(n_samples: int, seed: int, device):
    """Synthetic capture in HBM: a seeded tile of GFSK advertising packets (exponential gaps,
    random CFO/phase/length, SURVEY §8d cfg #2) repeated to n_samples, plus independent AWGN on
    every sample generated on the device. Returns (float32 tensor [2n], expected CRC-ok count,
    set of expected PDUs)."""
    import torch
    from snout_amd import synth
    tile, truth = synth.btle_capture(TILE, channel=37, seed=seed, noise=False)
    t = torch.from_numpy(tile.view(np.float32)).to(device)
    x = torch.empty(2 * n_samples, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(1000 + seed)
    reps = (n_samples + TILE - 1) // TILE
    for r in range(reps):
        lo = r * 2 * TILE
        hi = min(lo + 2 * TILE, 2 * n_samples)
        seg = x[lo:hi]
        torch.randn(seg.shape, generator=g, device=device, out=seg)
        seg.mul_(SIGMA).add_(t[:hi - lo])
    full = n_samples // TILE
    rem = n_samples - full * TILE
    expect = full * len(truth) + sum(1 for p in truth if p.sample_index + 1600 < rem)
    pdus = {p.payload for p in truth}
    torch.cuda.synchronize(device)
    return x, expect, pdus


def cpu_baseline(x_dev, n_sample: int, passes: int):
    """Time the CPU oracle (single thread) on the first n_sample samples of the workload."""
    from oracle import oracle_py
    oracle_py.lib()
    host = x_dev[:2 * n_sample].cpu().numpy()
    best = None
    n_pk = 0
    for _ in range(passes):
        t0 = time.perf_counter()
        pk, _ = oracle_py.btle_segment(host, channel=37, cap=max(1024, n_sample // 2048))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        n_pk = len(pk)
    return n_sample / best / 1e6, n_pk, host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=float, default=1e9, help="complex samples per GPU per step")
    ap.add_argument("--cpu-samples", type=float, default=2.5e8)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sync", action="store_true",
                    help="one segment at a time (no submit/collect pipelining); for profiling")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # one rank per GPU; SNOUT_BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box
    # with fewer GPUs than ranks (ranks then share devices; a debugging aid, not a measurement)
    backend = os.environ.get("SNOUT_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from snout_amd.rx import SnoutRx
    from snout_amd import dist as sdist

    n = int(args.samples)
    x, expect, pdus = make_workload(n, seed=2 + rank, device=device)
    rx = SnoutRx(proto=0, channel=37, device=local_rank)

    # Pipelined steps: up to three segments are in flight, so the next front-end kernel is already
    # queued when the previous one ends and the record D2H (copy stream) overlaps compute.  Every step's records are
    # in host memory (and gathered to rank 0) before the timed region ends.
    gather = sdist.AsyncRecordGather(device, width=80) if world > 1 else None

    def finish_one():
        pk = rx.collect(copy=False)
        if gather is not None:
            # RCCL gather of this step's records to rank 0, overlapped with the next step
            if len(gather.inflight) == 2:
                gather.finish(views=True)       # records are in rank 0's host memory; no host-side copy
            gather.start(pk, rx.last_records_device()[0])      # packed from the device copy: no upload
        return pk

    def run_steps(k):
        last = None
        if args.sync:
            for i in range(k):
                rx.submit(x, first_sample_index=rank * n)
                last = finish_one()
            return last
        for i in range(k):
            if gather is not None:
                gather.sync_uploads()       # result slots about to be reused have been read
            rx.submit(x, first_sample_index=rank * n)
            if i >= 2:                      # two segments stay queued behind the one being collected
                last = finish_one()
        for _ in range(min(k, 2)):
            last = finish_one()
        return last

    # prime the pipeline: first-use allocations of the three result slots and the HIP runtime's own
    # lazily grown pools (two ~7 ms stalls were measured around the 11th and 16th submit of a process)
    run_steps(24)
    if args.warmup:
        run_steps(args.warmup)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def drain():
        while gather is not None and gather.inflight:
            gather.finish(views=True)

    drain()
    fence()
    t0 = time.perf_counter()
    pk = run_steps(args.steps)
    drain()                 # every step's records are on rank 0 before the clock stops
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # dominant-kernel durations of the timed steps: HIP events recorded on the kernel's stream during
    # the timed region, read back only now (the library keeps the last 64 pairs)
    k_ms = rx.profile_history()[-min(args.steps, 64):]

    # correctness of the timed work: every generated packet decoded with a good CRC
    local = rx.process(x, first_sample_index=rank * n)
    n_ok = int(local["crc_ok"].sum())
    seen = {bytes(p["bytes"][:p["len"] - 3]) for p in local[:4096] if p["crc_ok"]}
    assert n_ok >= expect, f"rank {rank}: decoded {n_ok} CRC-ok packets, expected >= {expect}"
    assert seen <= pdus, "decoded a PDU that was never transmitted"
    prof = rx.profile()

    if rank == 0:
        total_samples = n * world * args.steps
        k_avg_ms = float(np.mean(k_ms))
        algo_bytes = 8.0 * n + 160.0 * len(local)
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if os.path.exists(tpath):       # PMC-derived HBM bytes per launch of the same workload
            tj = json.load(open(tpath))
            if tj.get("workload_samples") == n and tj.get("kernel") == prof.dominant_name:
                traffic = tj["traffic_bytes_per_launch"]
        out = {
            "metric": "complex-IQ Msamples/s through BTLE demod+correlate+decode",
            "value": total_samples / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: single-channel BTLE (ch37) GFSK demod + access-address "
                                   "correlate + dewhiten/CRC, %.3g cf32 samples per GPU resident in HBM"
                                   % n,
                       "samples_per_gpu": n, "packets_per_gpu": int(len(local)),
                       "decoded_pkts_per_s": len(local) * world * args.steps / dt,
                       "sharding": "segments per rank, RCCL gather of 160-B records" if world > 1
                                   else "single segment",
                       "stepping": "one segment at a time" if args.sync else
                                   "pipelined: record D2H of step i overlaps step i+1"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": "profiles/r1_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, "
                                           "gfx950-corrected)" if traffic else None,
                         "kernel": prof.dominant_name, "kernel_ms": k_avg_ms,
                         "algorithmic_bytes": algo_bytes},
        }
        if not args.no_cpu and world == 1:      # the CPU baseline is timed at N = 1 only
            ns = int(min(args.cpu_samples, n))
            v, n_pk, _ = cpu_baseline(x, ns, passes=3)
            out["cpu_baseline"] = {"value": v, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                   "sample": "first %.3g samples of the same workload, best of 3 "
                                             "passes, oracle/oracle_btle.c single thread" % ns}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
