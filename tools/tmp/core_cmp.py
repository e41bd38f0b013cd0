import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
tz, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
t = torch.from_numpy(np.ascontiguousarray(tz).view(np.float32)).cuda()
torch.manual_seed(5); x = t.repeat(16); x += 0.05 * torch.randn_like(x)
res = {}
CASES = [(2048, 512), (2048, 1024), (4096, 512), (4096, 1024), (4096, 2048), (8192, 2048), (1 << 22, 512)]
for core, warm in CASES:
    rx = SnoutRx(proto=1, n_channels=16, zb_core=core, zb_warmup=warm)
    r = rx.process(x, copy=True)
    ok = r[r["crc_ok"] == 1]
    key = lambda a: [(int(c), bytes(b[:l]), int(s)) for c, s, l, b in zip(a["channel"], a["sample_index"], a["len"], a["bytes"])]
    res[(core, warm)] = (len(r), key(ok), key(r[r["crc_ok"] == 0]))
    print(core, warm, "records", len(r), "crc ok", len(ok))
one = res[(1 << 22, 512)]
for core in CASES[:-1]:
    def diff(A, B):
        import collections
        d = collections.defaultdict(list)
        for c, b, s in B: d[(c, b)].append(s)
        return sum(1 for c, b, s in A if not any(abs(s - t) <= 8 for t in d.get((c, b), [])))
    print(core, "ok: missing vs one lane", diff(one[1], res[core][1]), "extra", diff(res[core][1], one[1]), "| bad: missing", diff(one[2], res[core][2]), "extra", diff(res[core][2], one[2]))
