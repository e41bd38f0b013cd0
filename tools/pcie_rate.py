"""PCIe-inclusive rate of the host-pointer entry point (snout_rx_process) for DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tile, _ = synth.btle_capture(1 << 22, seed=2)
x = np.tile(tile, 32)                     # 1.3e8 samples, 1 GiB, pageable host memory
rx = SnoutRx(proto=0, channel=37)
rx.process(x[:1 << 22])
for _ in range(3):
    t0 = time.perf_counter(); pk = rx.process(x, copy=False); dt = time.perf_counter() - t0
    print(f"host input: {x.size/dt/1e9:.2f} Gsamples/s ({8*x.size/dt/1e9:.1f} GB/s incl. H2D), {len(pk)} packets")
