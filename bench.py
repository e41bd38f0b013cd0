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

`--workload` selects one of the other SURVEY §8d configurations (cfg3: 40-channel BTLE wideband,
cfg4: 16-channel 802.15.4 wideband, zigbee1: single-channel 802.15.4) with the same contract; the
default, and the line the round is judged on, is cfg2.

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


WORKLOADS = {
    # name: (proto, n_channels, channel, default samples per GPU, record bytes gathered, metric, description)
    "cfg2": (0, 1, 37, 1e9, 80, "complex-IQ Msamples/s through BTLE demod+correlate+decode",
             "cfg2: single-channel BTLE (ch37) GFSK demod + access-address correlate + dewhiten/CRC"),
    "cfg3": (0, 40, 0, 40 * (1 << 22), 80,
             "wideband complex-IQ Msamples/s through 40-channel PFB + BTLE demod+correlate+decode",
             "cfg3: 80 Msps wideband -> 40-channel polyphase channelizer -> BTLE receive on every channel"),
    "cfg4": (1, 16, 0, 16 * (1 << 23), 160,
             "wideband complex-IQ Msamples/s through 16-channel PFB + 802.15.4 receive",
             "cfg4: 32 Msps wideband -> 16-channel polyphase channelizer -> 802.15.4 receive on every channel"),
    "zigbee1": (1, 1, 11, 1e9, 160, "complex-IQ Msamples/s through 802.15.4 O-QPSK receive",
                "single-channel 802.15.4 (ch11): discriminator + DC removal + M&M clock recovery + packet sink"),
}


def make_tile(workload: str, seed: int):
    """Noise-free host tile of the workload's traffic and its truth list."""
    from snout_amd import synth
    if workload == "cfg2":
        return synth.btle_capture(TILE, channel=37, seed=seed, noise=False)
    if workload == "zigbee1":
        return synth.zigbee_capture(TILE, channel=11, seed=seed, noise=False)
    if workload == "cfg3":
        return synth.wideband_capture(0, 40 * (1 << 16), seed=seed, sigma=0.0)
    if workload == "cfg4":
        return synth.wideband_capture(1, 16 * (1 << 17), seed=seed, sigma=0.0)
    raise SystemExit(f"unknown workload {workload}")


def make_workload(n_samples: int, seed: int, device, workload: str = "cfg2"):
    """Synthetic capture in HBM: a seeded tile of packets (exponential gaps, random CFO/phase/
    length) repeated to n_samples, plus independent AWGN on every sample generated on
    the device. Returns (float32 tensor [2n], expected CRC-ok count, set of expected PDUs)."""
    import torch
    tile, truth = make_tile(workload, seed)
    TILE = tile.size
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
    if workload == "cfg2":
        expect = full * len(truth) + sum(1 for p in truth if p.sample_index + 1600 < rem)
    else:
        # truth indices of the wideband tiles are at the channel rate and 802.15.4 frames are up to
        # 4256 samples long: count whole tiles only, and allow the frames a repetition cuts short
        # (cfg4: the synthetic 2 MHz raster makes adjacent 802.15.4 channels overlap spectrally,
        # DESIGN.md §6.7 -- with all 16 bins busy about half of the frames survive, on the oracle too)
        expect = int((0.4 if workload == "cfg4" else 0.9) * full * len(truth))
    pdus = {p.payload for p in truth}
    torch.cuda.synchronize(device)
    return x, expect, pdus


# bounded CPU sample per workload: ~10-30 s of single-thread oracle time for the three passes
CPU_SAMPLES = {"cfg2": 2.5e8, "cfg3": 40 * (1 << 21), "cfg4": 16 * (1 << 21), "zigbee1": 1 << 26}
CPU_SOURCE = {"cfg2": "oracle/oracle_btle.c", "cfg3": "oracle/oracle_pfb.c + oracle_btle.c",
              "cfg4": "oracle/oracle_pfb.c + oracle_zigbee.c", "zigbee1": "oracle/oracle_zigbee.c"}


def cpu_baseline(x_dev, n_sample: int, passes: int, workload: str = "cfg2"):
    """Time the CPU oracle (single thread) on the first n_sample samples of the workload."""
    from oracle import oracle_py
    oracle_py.lib()
    host = x_dev[:2 * n_sample].cpu().numpy()
    if host.dtype != np.float32:
        host = oracle_py.from_int(host)     # the oracle's definition of integer input (untimed)
    best = None
    n_pk = 0
    for _ in range(passes):
        t0 = time.perf_counter()
        if workload == "cfg2":
            pk, _ = oracle_py.btle_segment(host, channel=37, cap=max(1024, n_sample // 2048))
        elif workload == "zigbee1":
            pk = oracle_py.zigbee_segment(host, channel=11)
        else:
            pk = oracle_py.wideband_segment(host, proto=0 if workload == "cfg3" else 1)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        n_pk = len(pk)
    return n_sample / best / 1e6, n_pk, host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--samples", type=float, default=0,
                    help="complex input samples per GPU per step (default: the workload's size)")
    ap.add_argument("--format", choices=["cf32", "sc8", "sc16"], default="cf32",
                    help="input sample format resident in HBM (cf32 is BASELINE's; sc8 = HackRF / upstream "
                         "btle_rx int8 IQ, sc16 = USRP): the same capture quantised on the device")
    ap.add_argument("--cpu-samples", type=float, default=0,
                    help="samples of the CPU baseline leg (default: ~10-30 s of oracle time)")
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

    proto, n_ch, channel, n_default, rec_width, metric, descr = WORKLOADS[args.workload]
    n = int(args.samples or n_default)
    x, expect, pdus = make_workload(n, seed=2 + rank, device=device, workload=args.workload)
    fmt = {"cf32": 0, "sc8": 1, "sc16": 2}[args.format]
    if fmt:
        # what the SDR's ADC path would have delivered: full scale = 1.25 x the largest component
        bits = 7 if fmt == 1 else 15
        scale = float(1 << bits) / (1.25 * float(x.abs().max()))
        xi = torch.empty(x.shape, dtype=torch.int8 if fmt == 1 else torch.int16, device=device)
        step = 1 << 26
        for lo in range(0, x.numel(), step):
            xi[lo:lo + step] = (x[lo:lo + step] * scale).round_().clamp_(-(1 << bits), (1 << bits) - 1)
        x = xi
        del xi
        torch.cuda.empty_cache()
    rx = SnoutRx(proto=proto, channel=channel, n_channels=n_ch, device=local_rank, sample_format=fmt)

    # Pipelined steps: up to three segments are in flight, so the next front-end kernel is already
    # queued when the previous one ends and the record D2H (copy stream) overlaps compute.  Every step's records are
    # in host memory (and gathered to rank 0) before the timed region ends.
    gather = sdist.AsyncRecordGather(device, width=rec_width) if world > 1 else None

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
    fcs = 3 if proto == 0 else 0        # BTLE records carry PDU + CRC24, truth holds the PDU; 802.15.4: PSDU incl. FCS
    seen = {bytes(p["bytes"][:p["len"] - fcs]) for p in local[:4096] if p["crc_ok"]}
    assert n_ok >= expect, f"rank {rank}: decoded {n_ok} CRC-ok packets, expected >= {expect}"
    assert seen <= pdus, "decoded a PDU that was never transmitted"
    prof = rx.profile()

    if rank == 0:
        total_samples = n * world * args.steps
        k_avg_ms = float(np.mean(k_ms))
        algo_bytes = float(x.element_size() * 2) * n + 160.0 * len(local)
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if os.path.exists(tpath):       # PMC-derived HBM bytes per launch of the same workload
            tj = json.load(open(tpath))
            if (args.workload == "cfg2" and fmt == 0 and tj.get("workload_samples") == n
                    and tj.get("kernel") == prof.dominant_name):
                traffic = tj["traffic_bytes_per_launch"]
        out = {
            "metric": metric,
            "value": total_samples / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if fmt == 0 else ("i8->i32" if (fmt == 1 and args.workload == "cfg2") else
                                             args.format + "->f32"),
            "data": "synthetic",
            "config": {"workload": "%s, %.3g %s samples per GPU resident in HBM" % (descr, n, args.format),
                       "samples_per_gpu": n, "packets_per_gpu": int(len(local)),
                       "decoded_pkts_per_s": len(local) * world * args.steps / dt,
                       "sharding": "segments per rank, RCCL gather of 160-B records" if world > 1
                                   else "single segment",
                       "decoded_crc_ok_per_gpu": n_ok, "expected_crc_ok_per_gpu": expect,
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
            ns = int(min(args.cpu_samples or CPU_SAMPLES[args.workload], n))
            v, n_pk, _ = cpu_baseline(x, ns, passes=3, workload=args.workload)
            out["cpu_baseline"] = {"value": v, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                   "sample": "first %.3g samples of the same workload, best of 3 "
                                             "passes, %s single thread" % (ns, CPU_SOURCE[args.workload])}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
