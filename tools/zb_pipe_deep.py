import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tile, truth = synth.zigbee_capture(1 << 22, seed=4, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
for rep in (4, 24):
    x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
    n = x.numel() // 2
    rx = SnoutRx(proto=1, channel=11)
    for _ in range(3): rx.process(x, copy=False)
    K = 40
    for depth in (1, 2):            # the library keeps at most three segments in flight
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(K):
            rx.submit(x)
            if i >= depth: pk = rx.collect(copy=False)
        for _ in range(min(depth, K)): pk = rx.collect(copy=False)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"n={n:.3g} ahead={depth}: {1e3*(t2-t1)/K:.3f} ms/segment", flush=True)
    rx.close()
