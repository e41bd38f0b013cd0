"""Developer timing: Zigbee lane core 4096 vs 2048 on the 1-channel and 16-channel shapes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx

def run(name, rx, x, n_in, reps=6):
    for _ in range(2): pk = rx.process(x, copy=False)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); pk = rx.process(x, copy=False); ts.append(time.perf_counter() - t0)
    print(f"{name}: pkts={len(pk)} ok={int(pk['crc_ok'].sum())} wall={min(ts)*1e3:.3f} ms -> {n_in/min(ts)/1e9:.1f} Gs/s", flush=True)

tile, truth = synth.zigbee_capture(1 << 22, seed=4, noise=False)
t = torch.from_numpy(tile.view(np.float32)).cuda()
for rep in (4, 24, 238):
    x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
    for core in (4096, 2048):
        rx = SnoutRx(proto=1, channel=11, zb_core=core)
        run(f"1ch n={rep*tile.size:.3g} core={core} (expect {rep*len(truth)})", rx, x, rep * tile.size)
        del rx
    del x
tile, truth = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0)
t = torch.from_numpy(tile.view(np.float32)).cuda()
for rep in (8, 64, 152):
    x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
    for core in (4096, 2048):
        rx = SnoutRx(proto=1, n_channels=16, zb_core=core)
        run(f"16ch n_in={rep*tile.size:.3g} core={core}", rx, x, rep * tile.size)
        del rx
