"""Dev tool: where a tile's time goes in the 16-channel 802.15.4 channelizer (pfb_spec<16> built with -DSNOUT_MF_STAMPS).

    tools/pfb_variants.sh mfstamps:"-DSNOUT_MF_STAMPS" && python tools/mf_stamps16.py [samples] [fir waves, default 4]

Per wave, cycles per tile (median over the workgroups): FIR waves: fetch issue | FIR | staging | barrier wait;
FFT waves: work between barriers | barrier wait."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SNOUT_RX_LIB", os.path.join(ROOT, "build", "variants", "libsnout_rx_mfstamps.so"))
import numpy as np, torch
from snout_amd import synth, _ffi
from snout_amd.rx import SnoutRx

n_samples = float(sys.argv[1]) if len(sys.argv) > 1 else 3.2e8
nfir = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(os.environ.get("SNOUT_STAMPS_T", "128"))
tile, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=3, sigma=0.0)
x = torch.from_numpy(tile.view(np.float32)).cuda().repeat(max(1, int(n_samples) // tile.size))
torch.manual_seed(11)
x += 0.05 * torch.randn_like(x)
n = x.numel() // 2
rx = SnoutRx(proto=1, n_channels=16)
for _ in range(3):
    rx.process(x, copy=False)
hist = rx.profile_history()
lib = _ffi.load()
buf = np.zeros(256 * 16 * 8, dtype=np.uint64)
rc = lib.snout_debug_sp_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(buf.size))
assert rc == 0
s = buf.reshape(256, 16, 8).astype(np.float64)
tiles = (n - 256) // 8 // T / 256.0
print(f"n={n:.3g} kernel {np.mean(hist[-3:]):.3f} ms, {tiles:.0f} tiles of {T} per workgroup, clock {np.median(s[:, :, 6]) / 1e6:.2f} GHz")
print("wave: fetch  fir  stage  barrier | fft-work | total   (cycles per tile)")
for w in range(16):
    m = [float(np.median(s[:, w, j])) / tiles for j in range(8)]
    role = "FIR" if w < nfir else "FFT"
    print(f"  {role} {w:2d}: {m[0]:6.0f} {m[1]:6.0f} {m[4]:6.0f} {m[2]:6.0f} | {m[3]:6.0f} | {m[7]:6.0f}")
