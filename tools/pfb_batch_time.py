#!/usr/bin/env python3
"""Dev tool (GPU): duration of ONE channelizer launch over a batch of segments (HIP events around the kernel), for
different range splits: MIN_ITEM=48,200,1300 python tools/pfb_batch_time.py [proto] [count]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("CHILD"):
    import numpy as np, torch
    from snout_amd.rx import SnoutRx
    proto, count = int(sys.argv[1]), int(sys.argv[2])
    M = 40 if proto == 0 else 16
    seg = (1 << 24) + M * 2048
    x = torch.randn(2 * ((1 << 24) * count + seg), device="cuda") * 0.05
    rx = SnoutRx(proto=proto, n_channels=M, batch_segments=count)
    xs = [x[2 * k * (1 << 24): 2 * (k * (1 << 24) + seg)] for k in range(count)]
    firsts = [k * (1 << 24) // (M // 2) for k in range(count)]
    ms = []
    for it in range(6):
        rx.submit_batch(xs, firsts)
        rx.collect()
        ms.append(rx.profile().ms_dominant)
    print("proto %d count %d MIN_ITEM %s BLOCKS %s: kernel ms %s  (%.1f Gsamples/s)" % (proto, count, os.environ.get("SNOUT_PFB_MIN_ITEM"), os.environ.get("SNOUT_PFB_BLOCKS"),
          " ".join("%.3f" % m for m in ms), count * seg / min(ms) / 1e6))
else:
    proto = sys.argv[1] if len(sys.argv) > 1 else "0"
    count = sys.argv[2] if len(sys.argv) > 2 else "47"
    for mi in os.environ.get("MIN_ITEM", "48,100,200,400,1300").split(","):
        env = dict(os.environ, CHILD="1", SNOUT_PFB_MIN_ITEM=mi)
        subprocess.run([sys.executable, __file__, proto, count], env=env)
