"""Dev tool: zb_walk per-wave iteration counts and cycles.  Needs build/variants/libsnout_rx_walkstamps.so = the library
with zigbee.hip compiled with -DSNOUT_ZB_WALK_STAMPS (hipcc ... -DSNOUT_ZB_WALK_STAMPS -c zigbee.hip, linked with the
other objects of build/obj).

    python tools/walk_stamps.py [samples] [noise]

Round 3 (1e9 samples, 3 815 waves): 45 iterations per wave on traffic / 37 on noise, 13 800 / 7 100 cycles per iteration
for ~400 instructions; 37 of 64 lanes searching, 9 in a symbol, the rest done."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SNOUT_RX_LIB", os.path.join(ROOT, "build", "variants", "libsnout_rx_walkstamps.so"))
import numpy as np, torch
from snout_amd import synth, _ffi
from snout_amd.rx import SnoutRx
import bench
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
noise = len(sys.argv) > 2 and sys.argv[2] == "noise"
if noise:
    x = torch.randn(2 * n, device="cuda") * 0.3
else:
    tile, _ = bench.make_tile("zigbee1", seed=2)
    x = bench.resident_capture(tile, n, seed=2, device=torch.device("cuda", 0))
rx = SnoutRx(proto=1, channel=11)
for _ in range(2): pk = rx.process(x, copy=False)
lib = _ffi.load()
buf = np.zeros(8192 * 4, dtype=np.uint64)
assert lib.snout_debug_walk_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(buf.size)) == 0
s = buf.reshape(8192, 4).astype(np.float64)
s = s[s[:, 0] > 0]
print(f"n={n:.3g} {'noise' if noise else 'traffic'}: {len(pk)} records; waves sampled {len(s)}")
print(f"iterations per wave: mean {s[:,0].mean():.0f} p50 {np.median(s[:,0]):.0f} max {s[:,0].max():.0f}")
print(f"cycles per wave: mean {s[:,1].mean():.0f}; per iteration {s[:,1].sum()/s[:,0].sum():.0f}")
print(f"lanes per iteration: searching {s[:,2].sum()/s[:,0].sum():.1f}, in a symbol {s[:,3].sum()/s[:,0].sum():.1f} (of 64)")
