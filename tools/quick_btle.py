"""Quick on-GPU timing of the BTLE path at a few sizes (developer tool)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx

tile, truth = synth.btle_capture(1 << 22, seed=2, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
rx = SnoutRx(proto=0, channel=37)
for rep in (1, 16, 64, 238):
    n = rep * tile.size
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = t.repeat(rep)
    x += 0.05 * torch.randn(x.shape, device="cuda", generator=g)
    torch.cuda.synchronize()
    for _ in range(2):
        pk = rx.process(x)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pk = rx.process(x, copy=False); ts.append(time.perf_counter() - t0)
    p = rx.profile()
    print(f"n={n:>11d} pkts={len(pk)} ok={int(pk['crc_ok'].sum())} expect={rep*len(truth)} "
          f"wall={min(ts)*1e3:.3f} ms dev_total={p.ms_total:.3f} k1={p.ms_dominant:.3f} ms "
          f"k1 GB/s={8*n/p.ms_dominant/1e6:.0f} hits={p.n_hits}")
    del x
