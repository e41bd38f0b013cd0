"""A/B the btle_demod_corr variants in one process, interleaved rounds (developer tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx

n_rep = 238
tile, truth = synth.btle_capture(1 << 22, seed=2, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
x = t.repeat(n_rep); x += 0.05 * torch.randn_like(x)
torch.cuda.synchronize()
cfgs = [(1, 0), (2, 0), (3, 0), (4, 0)]
rxs = {}
for v, b in cfgs:
    os.environ["SNOUT_K1_DEPTH"] = str(v)
    rxs[(v, b)] = SnoutRx(proto=0, channel=37)
ref = None
res = {k: [] for k in rxs}
for rnd in range(6):
    for k, rx in rxs.items():
        pk = rx.process(x, copy=False)
        if ref is None: ref = pk.copy()
        assert len(pk) == len(ref) and np.array_equal(pk["sample_index"], ref["sample_index"]) and np.array_equal(pk["bytes"], ref["bytes"])
        if rnd: res[k].append(rx.profile().ms_dominant)
n = x.numel() // 2
for k, v in res.items():
    print(f"depth={k[0]}: "
          f"median {np.median(v):.4f} ms min {min(v):.4f} -> {8*n/np.median(v)/1e6:.0f} GB/s")
