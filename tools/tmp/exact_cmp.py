import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snout_amd import synth, _ffi
from snout_amd.rx import SnoutRx
lib = _ffi.load()
tz, _ = synth.wideband_capture(1, 16 * (1 << 17), seed=4, sigma=0.0, bins=range(0, 16, 2), max_len=100)
t = torch.from_numpy(np.ascontiguousarray(tz).view(np.float32)).cuda()
torch.manual_seed(5); x = t.repeat(16); x += 0.05 * torch.randn_like(x)
def run(core):
    rx = SnoutRx(proto=1, n_channels=16, zb_core=core)
    r = rx.process(x, copy=True)
    buf = (C.c_uint32 * 66)()
    lib.snout_debug_zb_passes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    rc = lib.snout_debug_zb_passes(rx._h, buf, 66)
    return r, list(buf)[:12], rc
one, _, _ = run(1 << 22)
for core in (2048, 4096):
    r, chg, rc = run(core)
    print("core", core, "records", len(r), "one lane", len(one), "identical:", r.tobytes() == one.tobytes(), "changed per pass", chg, rc)
    if len(r) == len(one) and r.tobytes() != one.tobytes():
        bad = [i for i in range(len(r)) if r[i].tobytes() != one[i].tobytes()]
        print("  differing records", len(bad), [(int(r[i]["sample_index"]) - int(one[i]["sample_index"])) for i in bad[:10]])
