"""Quick on-GPU timing of the Zigbee and wideband paths (developer tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx


def run(name, rx, x, n_in, reps=5):
    for _ in range(2):
        pk = rx.process(x, copy=False)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); pk = rx.process(x, copy=False); ts.append(time.perf_counter() - t0)
    p = rx.profile()
    print(f"{name}: n_in={n_in} pkts={len(pk)} ok={int(pk['crc_ok'].sum())} wall={min(ts)*1e3:.3f} ms "
          f"dev_total={p.ms_total:.3f} dom[{p.dominant_name}]={p.ms_dominant:.3f} ms "
          f"-> {n_in/min(ts)/1e9:.1f} Gsamples/s, {8*n_in/min(ts)/1e12:.2f} TB/s algorithmic", flush=True)


which = sys.argv[1:] or ["zb1", "btle40", "zb16"]
if "zb1" in which:
    tile, truth = synth.zigbee_capture(1 << 22, seed=4, noise=False)
    t = torch.from_numpy(tile.view(np.float32)).cuda()
    rep = 24
    x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
    for core in (4096, 2048):
        rx = SnoutRx(proto=1, channel=11, zb_core=core)
        run(f"zigbee 1ch core={core} (expect {rep*len(truth)})", rx, x, rep * tile.size)
        del rx
    del x
if "btle40" in which:
    tile, truth = synth.wideband_capture(0, 40 * (1 << 16), seed=3, sigma=0.0)
    t = torch.from_numpy(tile.view(np.float32)).cuda()
    rep = 64
    x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
    rx = SnoutRx(proto=0, n_channels=40)
    run(f"btle 40ch (expect ~{rep*len(truth)})", rx, x, rep * tile.size)
    del x, rx
if "zb16" in which:
    tile, truth = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0)
    t = torch.from_numpy(tile.view(np.float32)).cuda()
    rep = 64
    x = t.repeat(rep); x += 0.05 * torch.randn_like(x)
    rx = SnoutRx(proto=1, n_channels=16)
    run(f"zigbee 16ch (expect ~{rep*len(truth)})", rx, x, rep * tile.size)
