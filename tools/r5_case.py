"""Dev tool (gpurun): one single-channel 802.15.4 parity case of tools/fuzz_parity.py again, records that differ printed.
    python tools/r5_case.py n core warm seed gap cfo sigma [first]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snout_amd import synth
from snout_amd.rx import SnoutRx
from oracle import oracle_py as oracle
n, core, warm, seed = (int(v) for v in sys.argv[1:5])
gap, cfo, sigma = (float(v) for v in sys.argv[5:8])
first = int(sys.argv[8]) if len(sys.argv) > 8 else 0
x, _ = synth.zigbee_capture(n, seed=seed, mean_gap=gap, cfo_max_hz=cfo, sigma=sigma)
want = oracle.zigbee_segment(x, channel=11, core=core, warmup=warm, first_sample_index=first)
with SnoutRx(proto=1, channel=11, zb_core=core, zb_warmup=warm) as rx:
    got = rx.process(x, first_sample_index=first)
print("oracle", len(want), "GPU", len(got))
ws, gs = {r.tobytes() for r in want}, {r.tobytes() for r in got}
for name, rec, other in (("only oracle", want, gs), ("only GPU", got, ws)):
    for r in rec:
        if r.tobytes() not in other:
            print(name, "idx", int(r["sample_index"]) - first, "len", r["len"], "crc", r["crc_ok"], "lqi", r["lqi"], "flags", r["flags"], "lane", r["aux"], bytes(r["bytes"][:8]).hex())
