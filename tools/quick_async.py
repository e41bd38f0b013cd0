"""Sync vs pipelined stepping of the cfg #2 workload (developer tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx

tile, truth = synth.btle_capture(1 << 22, seed=2, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
x = t.repeat(238); x += 0.05 * torch.randn_like(x)
n = x.numel() // 2
torch.cuda.synchronize()
rx = SnoutRx(proto=0, channel=37)
K = 20
for _ in range(3): rx.process(x, copy=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
k = []
for _ in range(K):
    pk = rx.process(x, copy=False); k.append(rx.profile().ms_dominant)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"sync     : {dt/K*1e3:.3f} ms/step  k1 {np.mean(k):.3f} ms  pkts {len(pk)}")
for mode in ("pipelined", "pipelined-nostats"):
    rx.submit(x); rx.submit(x); rx.submit(x); rx.collect(copy=False); rx.collect(copy=False); rx.collect(copy=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = []
    for i in range(K):
        rx.submit(x)
        if i >= 2:
            pk = rx.collect(copy=False)
            if mode == "pipelined": k.append(rx.profile().ms_dominant)
    pk = rx.collect(copy=False); pk = rx.collect(copy=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    hist = rx.profile_history()[-K:]
    print(f"{mode:9s}: {dt/K*1e3:.3f} ms/step  k1 {np.mean(k) if k else 0:.3f} ms  hist-k1 {hist.mean():.3f}  pkts {len(pk)}")
