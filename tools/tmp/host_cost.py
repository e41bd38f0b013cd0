import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snout_amd import synth
from snout_amd.sharded import ShardedScan
from snout_amd.rx import SnoutRx

def tiled(tile, reps):
    t = torch.from_numpy(np.ascontiguousarray(tile).view(np.float32)).cuda()
    x = t.repeat(reps); x += 0.05 * torch.randn_like(x); return x
tb, _ = synth.wideband_capture(0, 40 * (1 << 16), seed=3, sigma=0.0)
tz, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
xb, xz = tiled(tb, 305), tiled(tz, 152)
for name, proto, M, x in (("btle40", 0, 40, xb), ("zigbee16", 1, 16, xz)):
    n = x.numel() // 2
    for H in (1, 2, 4):
        sc = ShardedScan(proto, n_channels=M, seg_len=1 << 24, handles=H)
        src = lambda a, b: x[2 * a:2 * b]
        # instrument submit / collect
        tsub = tcol = 0.0; nsub = 0
        for rx in sc.rxs:
            osub, ocol = rx.submit, rx.collect
            def mk(osub=osub, ocol=ocol):
                def sub(*a, **k):
                    global tsub, nsub
                    t = time.perf_counter(); r = osub(*a, **k); tsub += time.perf_counter() - t; nsub += 1; return r
                def col(*a, **k):
                    global tcol
                    t = time.perf_counter(); r = ocol(*a, **k); tcol += time.perf_counter() - t; return r
                return sub, col
            rx.submit, rx.collect = mk()
        for rep in range(3):
            tsub = tcol = 0.0; nsub = 0
            st = {}
            sc.run(n, src, stats=st)
        print(f"{name} handles {H}: loop {1e3*st['device_s']:.2f} ms for {nsub} segments; host in submit {1e3*tsub:.2f} ms ({1e6*tsub/nsub:.0f} us each), in collect (incl. waiting) {1e3*tcol:.2f} ms")
        # single-segment latency, synchronous
        rx = SnoutRx(proto=proto, n_channels=M)
        seg = x[:2 * (1 << 24)]
        for _ in range(3): rx.process(seg, copy=False)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): rx.process(seg, copy=False)
        torch.cuda.synchronize(); print(f"   one 2^24 segment, synchronous call: {1e5*(time.perf_counter()-t):.0f} us")
        sc.close(); rx.close()
