import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd.rx import SnoutRx
x = torch.randn(2 * 40 * (1 << 22), device="cuda")
n = x.numel() // 2
for ab in (0, 8, 12, 14, 15, 1, 2, 4, 6):
    os.environ["SNOUT_PFB_ABLATE"] = str(ab)
    rx = SnoutRx(proto=0, n_channels=40)
    for _ in range(3): rx.process(x, copy=False)
    print(f"ablate={ab:2d} (noFIR={ab&1} no3a={(ab>>1)&1} no3b={(ab>>2)&1} noStore={(ab>>3)&1}): pfb {rx.profile().ms_dominant:.3f} ms -> {n/rx.profile().ms_dominant/1e6:.0f} Gs/s", flush=True)
    rx.close()
