"""Developer timing: Zigbee 16-channel wideband, one 2^24-input-sample segment (cfg #5 shape)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tile, truth = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
t = torch.from_numpy(tile.view(np.float32)).cuda()
x = t.repeat(8); x += 0.05 * torch.randn_like(x)
rx = SnoutRx(proto=1, n_channels=16, zb_core=int(os.environ.get("ZB_CORE", "0")), zb_warmup=int(os.environ.get("ZB_WARM", "0")))
for _ in range(3):
    t0 = time.perf_counter(); pk = rx.process(x, copy=False); dt = time.perf_counter() - t0
p = rx.profile()
print(f"n_in={x.numel()//2} wall={dt*1e3:.3f} ms dev_total={p.ms_total:.3f} pkts={len(pk)}")
