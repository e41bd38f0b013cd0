"""Developer timing: Zigbee kernels on noise only (no frames): the search-only cost of zb_walk."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd.rx import SnoutRx
x = torch.randn(2 * 100663296, device="cuda") * 0.3
rx = SnoutRx(proto=1, channel=11)
for _ in range(2): pk = rx.process(x, copy=False)
print(rx.profile().ms_dominant, len(pk))
