"""Dev tool: 802.15.4 chain time against the lane core around the sizes where the lanes fill whole rounds of the GPU
(256 CUs x 12 zb_mm waves x 64 lanes = 196 608 lanes per round).

    python tools/zb_rounds.py            # single channel 1e9 samples, and 16 channels x 4e7 (cfg #4's channel samples)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
L = 256 * 12 * 64
tile, truth = synth.zigbee_capture(1 << 22, seed=4, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
x = t.repeat(238); x += 0.05 * torch.randn_like(x)          # 9.98e8 samples
n = x.numel() // 2
for core in (2560, 4096, 5056, 5120, 5184, 5632, 6144):
    rx = SnoutRx(proto=1, channel=11, zb_core=core, zb_warmup=512)
    ks = []
    for _ in range(3):
        pk = rx.process(x, copy=False); ks.append(rx.profile().ms_dominant)
    lanes = -(-n // core)
    print(f"1 channel n={n:.3g} core={core}: chain {min(ks):.3f} ms, {lanes} lanes = {lanes / L:.2f} rounds, pkts={len(pk)} ok={int(pk['crc_ok'].sum())}", flush=True)
    rx.close()
