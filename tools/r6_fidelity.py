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
shapes = [(6144, 1024), (6144, 2048), (6144, 3072), (6144, 4096), (4096, 2048), (4096, 3072), (8192, 3072), (8192, 4096), (16384, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("/")) for a in sys.argv[1:]]


def timed(workload, x, core, warm, n_ch, channel=0):
    with SnoutRx(proto=1, n_channels=n_ch, channel=channel, device=0, zb_core=core, zb_warmup=warm) as rx:
        def loop(m):
            for i in range(m):
                rx.submit(x)
                if i >= 2:
                    rx.collect(copy=False)
            for _ in range(min(m, 2)):
                rx.collect(copy=False)
        loop(12)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(20)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e3


tot = {sh: [0, 0, 0, 0] for sh in shapes}
ms = {}
for seed in (2, 4):                     # the tile sets of bench.py's cfg #4 and cfg #5 captures
    tiles, truths = bench.make_tiles("cfg4", seed, dev)
    x = bench.resident_capture(tiles, int(3.2e8), seed=seed, device=dev)
    del tiles
    res = bench.lost_vs_sequential(x, "cfg4", 1 << 26, dev, 0, shapes=shapes)
    print("tile set %d: sequential frames %d (distinct %d), one lane per channel %.1f Msamples/s" % (
        seed, res["sequential_frames"], res["distinct_sequential_frames"], res["fidelity_modes"]["one lane per channel"]["Msamples_per_s"]))
    for sh in shapes:
        f = res["fidelity_modes"]["%d / %d" % sh]
        tot[sh][0] += f["lost"]; tot[sh][1] += f["extra"]; tot[sh][2] += f["repaired"]; tot[sh][3] += res["sequential_frames"]
        if seed == 2:
            ms[sh] = timed("cfg4", x, sh[0], sh[1], 16)
    del x
    torch.cuda.empty_cache()
t1, _ = bench.make_tiles("zigbee1", 2, dev)
x1 = bench.resident_capture(t1, int(1e9), seed=2, device=dev)
print("core / warm-up     lost   extra   lost %  extra %  repaired   cfg4 ms per step   single channel 1e9 ms per step")
for sh in shapes:
    lo, ex, rep, nseq = tot[sh]
    z1 = timed("zigbee1", x1, sh[0], sh[1], 1, 11)
    print("%6d / %-6d %6d %7d %8.2f %8.2f %9d   %8.3f   %8.3f" % (sh[0], sh[1], lo, ex, 100.0 * lo / nseq, 100.0 * ex / nseq, rep, ms[sh], z1))
