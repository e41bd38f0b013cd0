#!/usr/bin/env python3
"""Dev tool (gpurun), round 6: 802.15.4 lane shapes on cfg #4's traffic as bench.py builds it (32 distinct tiles, 2^26
samples): frames lost / extra against ONE sequential lane per channel, and the pipelined step time of the 3.2e8-sample
workload, per (core, warm-up).  -> profiles/r6_fidelity.md"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from snout_amd.rx import SnoutRx                               # noqa: E402

dev = torch.device("cuda", 0)
shapes = [(6144, 1024), (6144, 2048), (6144, 3072), (8192, 1024), (8192, 2048), (8192, 4096), (12288, 4096), (16384, 4096), (16384, 8192)]
tiles, truths = bench.make_tiles("cfg4", 2, dev)
n = int(3.2e8)
x = bench.resident_capture(tiles, n, seed=2, device=dev)
del tiles
res = bench.lost_vs_sequential(x, "cfg4", 1 << 26, dev, 0, shapes=shapes)
print("sequential frames %d (distinct %d), one lane per channel %.1f Msamples/s" % (
    res["sequential_frames"], res["distinct_sequential_frames"], res["fidelity_modes"]["one lane per channel"]["Msamples_per_s"]))
print("core / warm-up     lost   extra   lost %  extra %  repaired   ms per step (3.2e8 samples, pipelined)")
for core, warm in shapes:
    with SnoutRx(proto=1, n_channels=16, device=0, zb_core=core, zb_warmup=warm) as rx:
        def loop(m):
            for i in range(m):
                rx.submit(x)
                if i >= 2:
                    rx.collect(copy=False)
            for _ in range(min(m, 2)):
                rx.collect(copy=False)
        loop(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(20)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
    f = res["fidelity_modes"]["%d / %d" % (core, warm)]
    print("%6d / %-6d %6d %7d %8.2f %8.2f %9d   %.3f" % (core, warm, f["lost"], f["extra"], 100 * f["frac_lost"], 100 * f["frac_extra"], f["repaired"], ms))
