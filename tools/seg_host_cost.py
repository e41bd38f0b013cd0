"""Where the host time of a small pipelined segment goes (developer tool): 2^24-sample wideband
BTLE segments through submit / collect, wall time of the two C calls against the whole loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tb, _ = synth.wideband_capture(0, 40 * (1 << 16), seed=3, sigma=0.0)
t = torch.from_numpy(np.ascontiguousarray(tb).view(np.float32)).cuda()
x = t.repeat(8); x += 0.05 * torch.randn_like(x)            # 2.1e7 input samples
n = x.numel() // 2
rx = SnoutRx(proto=0, n_channels=40)
for _ in range(6):
    rx.submit(x); rx.collect(copy=False)
K = 200
ts = tc = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter(); rx.submit(x); b = time.perf_counter(); ts += b - a
    if i >= 2:
        a = time.perf_counter(); pk = rx.collect(copy=False); b = time.perf_counter(); tc += b - a
for _ in range(2): pk = rx.collect(copy=False)
torch.cuda.synchronize(); t1 = time.perf_counter()
p = rx.profile()
print(f"n={n:.3g}: {1e3*(t1-t0)/K:.3f} ms/segment; submit {1e3*ts/K:.3f} ms, collect {1e3*tc/K:.3f} ms (waits included); "
      f"kernels of one segment {p.ms_total:.3f} ms (dominant {p.ms_dominant:.3f})")
