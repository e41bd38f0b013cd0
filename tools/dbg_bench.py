import os, sys, time, json, faulthandler
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from snout_amd.rx import SnoutRx
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n = int(1e9)
x, expect, pdus = bench.make_workload(n, seed=2, device=dev)
rx = SnoutRx(proto=0, channel=37, device=0)
def run(k, log=None):
    for i in range(k):
        t0 = time.perf_counter(); rx.submit(x); t1 = time.perf_counter()
        if i >= 2:
            pk = rx.collect(copy=False)
        t2 = time.perf_counter()
        if log is not None: log.append((t1 - t0, t2 - t1))
    rx.collect(copy=False)
    return rx.collect(copy=False)
run(4)
run(3)
for rep in range(3):
    log = []
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(20, log); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    a = np.array(log) * 1e3
    print("   slow calls:", [(i, round(v[0],2), round(v[1],2)) for i, v in enumerate(a) if v[0] > 0.5 or v[1] > 2.0])
    print(f"rep{rep}: {dt/20*1e3:.3f} ms/step; submit mean {a[:,0].mean():.3f} max {a[:,0].max():.3f}; collect mean {a[1:,1].mean():.3f} max {a[1:,1].max():.3f}; k1 {rx.profile_history()[-20:].mean():.3f}")
print(np.round(a[:8], 3))
