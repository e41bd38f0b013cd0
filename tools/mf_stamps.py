"""Dev tool: where a tile's time goes in the matrix-pipe channelizer (pfb_mfma.hip built with -DSNOUT_MF_STAMPS).

    tools/pfb_variants.sh mfstamps:"-DSNOUT_MF_STAMPS" && python tools/mf_stamps.py [samples]

Per role (FIR waves 0-7, FFT waves 8-15), cycles per tile and wave, median over the workgroups:
  FIR: staging + fetch issue | the MFMA chains | waiting at the barrier
  FFT: work between barriers | waiting at the barrier"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SNOUT_RX_LIB", os.path.join(ROOT, "build", "variants", "libsnout_rx_mfstamps.so"))
import numpy as np, torch
from snout_amd import synth, _ffi
from snout_amd.rx import SnoutRx

n_samples = float(sys.argv[1]) if len(sys.argv) > 1 else 8e8
tile, _ = synth.wideband_capture(0, 40 * (1 << 16), seed=3, sigma=0.0)
x = torch.from_numpy(tile.view(np.float32)).cuda().repeat(max(1, int(n_samples) // tile.size))
torch.manual_seed(11)
x += 0.05 * torch.randn_like(x)
n = x.numel() // 2
rx = SnoutRx(proto=0, n_channels=40)
for _ in range(3):
    rx.process(x, copy=False)
k = rx.profile_history()[-3:]
lib = _ffi.load()
buf = np.zeros(256 * 16 * 8, dtype=np.uint64)
fn = lib.snout_debug_mf_stamps if os.environ.get('SNOUT_PFB_IMPL', 'spec') in ('mfma', 'spec16') else lib.snout_debug_sp_stamps
rc = fn(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(buf.size))
assert rc == 0
s = buf.reshape(256, 16, 8).astype(np.float64)
tiles = (n - 640) // 20 // 128 / 256.0
print(f"n={n:.3g} kernel {k.mean():.3f} ms, {tiles:.0f} tiles per workgroup, clock {np.median(s[:, :, 6]) / 1e6:.2f} GHz")
fir, fft = s[:, :8, :], s[:, 8:, :]
med = lambda a: float(np.median(a)) / tiles
print("(front-end waves 0-7: FIR = slots 0 1 2 = before / in / after the FIR; staging waves: slot 0 = fetch+stage incl. load wait, 2 = barrier)")
print(f"FIR waves, cycles per tile: stage+fetch {med(fir[:, :, 0]):.0f}  mfma chains {med(fir[:, :, 1]):.0f}  barrier {med(fir[:, :, 2]):.0f}  total {med(fir[:, :, 7]):.0f}")
print(f"FFT waves, cycles per tile: work {med(fft[:, :, 3]):.0f}  barrier {med(fft[:, :, 2]):.0f}  total {med(fft[:, :, 7]):.0f}")
for w in range(16):
    print(f"  wave {w:2d}: " + " ".join(f"{float(np.median(s[:, w, j])) / tiles:7.0f}" for j in (0, 1, 4, 2, 3, 7)))
