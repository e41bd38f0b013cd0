"""Lane core-size sweep of the single-channel Zigbee path (developer tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tile, truth = synth.zigbee_capture(1 << 22, seed=4, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
x = t.repeat(24); x += 0.05 * torch.randn_like(x)
for core in (2048, 4096, 8192, 16384):
    rx = SnoutRx(proto=1, channel=11, zb_core=core)
    for _ in range(2): pk = rx.process(x, copy=False)
    print(f"core={core}: {rx.profile().ms_dominant:.3f} ms pkts={len(pk)} ok={int(pk['crc_ok'].sum())} (sent {24*len(truth)})", flush=True)
    rx.close()
