"""Developer timing: Zigbee single channel at 1e9 samples and 16-channel wideband."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tile, truth = synth.zigbee_capture(1 << 22, seed=4, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
rep = int(os.environ.get("REP", "238"))
x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
rx = SnoutRx(proto=1, channel=11, zb_core=int(os.environ.get("CORE", "0")), zb_warmup=int(os.environ.get("WARM", "0")))
for _ in range(3):
    t0 = time.perf_counter(); pk = rx.process(x, copy=False); dt = time.perf_counter() - t0
print(f"n={rep*tile.size:.3g} wall={dt*1e3:.2f} ms dom={rx.profile().ms_dominant:.2f} ms pkts={len(pk)} ok={int(pk['crc_ok'].sum())} expect={rep*len(truth)}"
      f" -> {rep*tile.size/dt/1e9:.1f} Gsamples/s")
