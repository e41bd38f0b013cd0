"""Dev tool: per-phase cycles of the channelizer's tile loop from the -DSNOUT_PFB_STAMPS build
(tools/pfb_variants.sh stamps:"-DSNOUT_PFB_STAMPS"; run through gpurun)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SNOUT_RX_LIB"] = os.path.join(ROOT, "build", "variants", "libsnout_rx_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamps"))
from snout_amd import synth, _ffi
from snout_amd.rx import SnoutRx
tile, _ = synth.wideband_capture(0, 40 * (1 << 16), seed=3, sigma=0.0)
x = torch.from_numpy(tile.view(np.float32)).cuda().repeat(305)
x += 0.05 * torch.randn_like(x)
rx = SnoutRx(proto=0, n_channels=40)
for _ in range(3):
    rx.process(x, copy=False)
lib = _ffi.load()
buf = (C.c_uint64 * (768 * 5 * 8))()
assert lib.snout_debug_pfb_stamps(buf, 768 * 5 * 8) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(768, 5, 8).astype(np.float64)
tiles = (x.numel() // 2 // 1280) / 768
names = ["stage (loads wait, LDS writes)", "barrier after stage", "FIR", "barrier after FIR", "pass 3a", "barrier after 3a", "pass 3b + slicer"]
print("kernel ms", rx.profile_history()[-1], "tiles per workgroup %.0f" % tiles)
tot = a[:, :, :7].sum(axis=2).mean()
for k, nm in enumerate(names):
    per = a[:, :, k] / tiles
    print("%-32s mean %7.0f cycles/tile  (waves 0-3 %7.0f, wave 4 %7.0f)  %4.1f %%" % (nm, per.mean(), per[:, :4].mean(), per[:, 4].mean(), 100 * a[:, :, k].mean() / tot))
print("sum per tile %.0f cycles (s_memtime ticks)" % (tot / tiles))
print("shader clock from s_memtime / s_memrealtime: median %.0f MHz (min %.0f, max %.0f)" % (np.median(a[:, :, 7]) / 1e3, a[:, :, 7].min() / 1e3, a[:, :, 7].max() / 1e3))
tb = (C.c_uint64 * (768 * 4))()
assert lib.snout_debug_pfb_times(tb, 768 * 4) == 0
tt = np.frombuffer(tb, dtype=np.uint64).reshape(768, 4).astype(np.int64)
t0 = tt[:, 0].min()
start, end = (tt[:, 0] - t0) / 100.0, (tt[:, 1] - t0) / 100.0           # microseconds (100 MHz)
print("workgroup start (us): min %.1f median %.1f max %.1f;  end: min %.1f median %.1f max %.1f" % (start.min(), np.median(start), start.max(), end.min(), np.median(end), end.max()))
late = start > 0.25 * end.max()
print("workgroups that start late (after 25 %% of the kernel): %d of 768; their run time median %.0f us vs early ones %.0f us" % (late.sum(), np.median((end - start)[late]) if late.any() else 0, np.median((end - start)[~late])))
hw = tt[:, 3]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = tt[:, 2] * 1000 + se * 100 + sh * 16 + cu
import collections
cnt = collections.Counter(key[~late].tolist())
print("early workgroups per (xcc, se, sh, cu): histogram of counts", collections.Counter(cnt.values()))
# residency per CU: maximum number of overlapping workgroups, and how long workgroups run by CU load
byk = collections.defaultdict(list)
for i in range(768):
    if tt[i, 1] > tt[i, 0]:
        byk[int(key[i])].append(i)
load = {}
for k, idx in byk.items():
    ev = sorted([(start[i], 1) for i in idx] + [(end[i], -1) for i in idx])
    cur = best = 0
    for _, d in ev:
        cur += d; best = max(best, cur)
    load[k] = best
print("CUs seen %d; max overlapping workgroups per CU: histogram" % len(load), sorted(collections.Counter(load.values()).items()))
for L in sorted(set(load.values())):
    idx = [i for k, v in byk.items() if load[k] == L for i in v]
    print("  CUs with %d resident: %d workgroups, run time median %.0f us, last end %.0f us" % (L, len(idx), np.median((end - start)[idx]), end[idx].max()))
ok = tt[:, 1] > tt[:, 0]
dur = (end - start)[ok]
e = end[ok] - start[ok].min()
print("workgroup run time (us): min %.0f p10 %.0f median %.0f p90 %.0f max %.0f; last end - median end = %.0f us (%.1f %% of the kernel)" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), e.max() - np.median(e), 100 * (e.max() - np.median(e)) / e.max()))
