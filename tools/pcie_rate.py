"""PCIe-inclusive rate of the host-pointer entry point (snout_rx_process) for DESIGN.md:
the same capture as cf32 and re-quantised to sc8 / sc16 (pageable host memory)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tile, _ = synth.btle_capture(1 << 22, seed=2)
x = np.tile(tile, 32)                     # 1.3e8 samples, 1 GiB as cf32
for name, fmt in (("cf32", 0), ("sc8", 1), ("sc16", 2)):
    a = x if fmt == 0 else synth.quantize(x, fmt)
    n = x.size
    rx = SnoutRx(proto=0, channel=37, sample_format=fmt)
    rx.process(a[:(1 << 22) * (1 if fmt == 0 else 2)])
    for _ in range(3):
        t0 = time.perf_counter(); pk = rx.process(a, copy=False); dt = time.perf_counter() - t0
    print(f"host input {name}: {n/dt/1e9:.2f} Gsamples/s ({a.nbytes/dt/1e9:.1f} GB/s incl. H2D), {len(pk)} packets", flush=True)
    rx.close()
