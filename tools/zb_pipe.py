"""Developer timing: Zigbee through submit/collect with segments in flight (tail of segment i on the
tail stream overlaps the front end of segment i+1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
wide = os.environ.get("WIDE", "0") == "1"
if wide:
    tile, truth = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
    rep = int(os.environ.get("REP", "64"))
else:
    tile, truth = synth.zigbee_capture(1 << 22, seed=4, noise=False)
    rep = int(os.environ.get("REP", "24"))
t = torch.from_numpy(tile.view(np.float32)).cuda()
x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
n = x.numel() // 2
rx = SnoutRx(proto=1, channel=11, n_channels=16 if wide else 1)
for _ in range(3): rx.process(x, copy=False)
K = int(os.environ.get("K", "40"))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): pk = rx.process(x, copy=False)
for i in range(4):                      # first use of the second work set and of the speculative copy
    rx.submit(x)
    if i >= 2: rx.collect(copy=False)
for _ in range(2): rx.collect(copy=False)
torch.cuda.synchronize(); t1 = time.perf_counter()
for i in range(K):
    rx.submit(x)
    if i >= 2: pk = rx.collect(copy=False)
for _ in range(2): pk = rx.collect(copy=False)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"n={n:.3g} one at a time {1e3*(t1-t0)/K:.3f} ms/segment ({n*K/(t1-t0)/1e9:.1f} Gs/s); "
      f"three in flight {1e3*(t2-t1)/K:.3f} ms/segment ({n*K/(t2-t1)/1e9:.1f} Gs/s); pkts={len(pk)}")
