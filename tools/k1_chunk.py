import os, sys, shutil
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth, _ffi
if len(sys.argv) > 1:
    _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), sys.argv[1])
from snout_amd.rx import SnoutRx
tile, truth = synth.btle_capture(1 << 22, seed=2, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
x = t.repeat(238); x += 0.05 * torch.randn_like(x)
rx = SnoutRx(proto=0, channel=37)
for _ in range(4): pk = rx.process(x, copy=False)
k = []
for _ in range(12):
    pk = rx.process(x, copy=False); k.append(rx.profile().ms_dominant)
print(sys.argv[1:] or "default", f"k1 median {np.median(k):.4f} min {min(k):.4f} pkts {len(pk)} ok {int(pk['crc_ok'].sum())}")
