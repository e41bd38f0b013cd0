"""Developer timing of cfg #5 on one GPU: BTLE 40-channel and Zigbee 16-channel wideband scans,
segments of 2^24 input samples, run one after the other and concurrently (one stream each)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.sharded import ShardedScan, run_concurrent


def tiled(tile, reps):
    t = torch.from_numpy(np.ascontiguousarray(tile).view(np.float32)).cuda()
    x = t.repeat(reps)
    x += 0.05 * torch.randn_like(x)
    return x


tb, _ = synth.wideband_capture(0, 40 * (1 << 16), seed=3, sigma=0.0)
tz, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
xb, xz = tiled(tb, 305), tiled(tz, 152)            # 10 s of each band: 8.0e8 and 3.2e8 input samples
nb, nz = xb.numel() // 2, xz.numel() // 2
srcb = lambda a, b: xb[2 * a:2 * b]
srcz = lambda a, b: xz[2 * a:2 * b]
SEG = int(os.environ.get("SEG", str(1 << 24)))
HB, HZ = int(os.environ.get("HB", "1")), int(os.environ.get("HZ", "2"))
sb = ShardedScan(0, n_channels=40, seg_len=SEG, handles=HB, batch=int(os.environ.get("BB", "4")))
sz = ShardedScan(1, n_channels=16, seg_len=SEG, handles=HZ, batch=int(os.environ.get("BZ", "4")), zb_core=int(os.environ.get("ZB_CORE", "0")), zb_warmup=int(os.environ.get("ZB_WARM", "0")))
for rep in range(2):
    sa, sb2, sc = {}, {}, {}
    a = sb.run(nb, srcb, stats=sa)
    b = sz.run(nz, srcz, stats=sb2)
    c, d = run_concurrent([sb, sz], [nb, nz], [srcb, srcz], stats=sc)
print(f"btle40 alone: device {1e3*sa['device_s']:.1f} ms ({nb/sa['device_s']/1e9:.1f} Gs/s) + host post {1e3*sa['post_s']:.1f} ms, {len(a)} pkts")
print(f"zigbee16 alone: device {1e3*sb2['device_s']:.1f} ms ({nz/sb2['device_s']/1e9:.1f} Gs/s) + host post {1e3*sb2['post_s']:.1f} ms, {len(b)} pkts")
print(f"both concurrently: device {1e3*sc['device_s']:.1f} ms ({(nb+nz)/sc['device_s']/1e9:.1f} Gs/s) + host post {1e3*sc['post_s']:.1f} ms; "
      f"same records: {np.array_equal(a, c) and np.array_equal(b, d)}")
