"""Developer timing: host-side cost of the per-step record gather (nccl process group of one rank):
the bench loop with and without AsyncRecordGather."""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import bench
from snout_amd.rx import SnoutRx
from snout_amd import dist as sdist
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
tile, _ = bench.make_tile("cfg2", seed=2)
x = bench.resident_capture(tile, int(1e9), seed=2, device=dev)
rx = SnoutRx(proto=0, channel=37, device=0)


def loop(k, gather):
    for i in range(k):
        if gather is not None: gather.sync_uploads()
        rx.submit(x)
        if i >= 2:
            pk = rx.collect(copy=False)
            if gather is not None:
                if len(gather.inflight) == 2: gather.finish(views=True)
                gather.start(pk, rx.last_records_device()[0])
    for _ in range(2):
        pk = rx.collect(copy=False)
        if gather is not None:
            if len(gather.inflight) == 2: gather.finish(views=True)
            gather.start(pk, rx.last_records_device()[0])
    while gather is not None and gather.inflight: gather.finish(views=True)


for name, g in (("no gather", None), ("gather", sdist.AsyncRecordGather(dev, width=80))):
    loop(24, g)
    torch.cuda.synchronize(); t0 = time.perf_counter(); loop(20, g); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step, kernel {np.mean(rx.profile_history()[-20:]):.3f} ms", flush=True)
    if g is not None:
        pk = rx.process(x, copy=False)
        t0 = time.perf_counter()
        for _ in range(10): g.start(pk, rx.last_records_device()[0]); g.finish(views=True)
        print(f"start+finish alone: {(time.perf_counter()-t0)/10*1e3:.3f} ms for {len(pk)} records", flush=True)
dist.destroy_process_group()
