"""Multi-GPU path on CPU: segment sharding, record dedup, and the record gather over a real
world_size-2 gloo process group."""
import os
import socket

import numpy as np
import pytest

from snout_amd import dist as sdist
from snout_amd._ffi import PKT_DTYPE


def test_shard_segments_cover_everything_once():
    n, seg, ov = 1000, 128, 40
    for world in (1, 2, 3, 8):
        segs = [sdist.shard_segments(n, seg, ov, r, world) for r in range(world)]
        flat = sorted(s for rank in segs for s in rank)
        assert [s[0] for s in flat] == list(range(0, n, seg))
        assert all(b - a <= seg + ov and b <= n for a, b in flat)
        assert all(flat[i][1] >= min(flat[i + 1][0] + ov, n) for i in range(len(flat) - 1))
        counts = [len(r) for r in segs]
        assert max(counts) - min(counts) <= 1


def _recs(keys):
    r = np.zeros(len(keys), dtype=PKT_DTYPE)
    for i, (p, c, s) in enumerate(keys):
        r[i]["proto"], r[i]["channel"], r[i]["sample_index"] = p, c, s
        r[i]["bytes"][0] = i & 0xFF
    return r


def test_dedup_records():
    r = _recs([(0, 37, 500), (0, 37, 100), (1, 11, 100), (0, 37, 500), (0, 38, 100), (0, 37, 100)])
    d = sdist.dedup_records(r)
    assert [(int(x["proto"]), int(x["channel"]), int(x["sample_index"])) for x in d] == \
        [(0, 37, 100), (0, 37, 500), (0, 38, 100), (1, 11, 100)]
    assert sdist.dedup_records(r[:0]).size == 0


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 3 if rank == 0 else 5
        rec = _recs([(0, 37, 1000 * rank + i) for i in range(n)])
        out = sdist.gather_records(rec)
        empty = sdist.gather_records(rec[:0] if rank == 1 else rec[:1])
        # pipelined gatherer: two in flight, compact 80-byte wire records, growing capacity
        g = sdist.AsyncRecordGather(width=80)
        seq = []
        for step in range(4):
            k = 2 + step + rank if step < 3 else 700      # within the agreed capacity (1.25 * 3 + 1024)
            r = _recs([(0, 37, 10000 * step + 100 * rank + i) for i in range(k)])
            if step == 3:
                while g.inflight:
                    seq.append(g.finish())
            elif len(g.inflight) == 2:
                seq.append(g.finish())
            g.start(r)
        while g.inflight:
            seq.append(g.finish())
        if rank == 0:
            q.put((out["sample_index"].tolist(), empty["sample_index"].tolist(),
                   [s["sample_index"].tolist() for s in seq], int(seq[0].dtype.itemsize)))
        else:
            assert out is None and empty is None and all(s is None for s in seq)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_records_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, empty, seq, width = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == [0, 1, 2, 1000, 1001, 1002, 1003, 1004]
    assert empty == [0]
    assert width == 80 and len(seq) == 4
    for step in range(3):
        want = [10000 * step + i for i in range(2 + step)] + [10000 * step + 100 + i for i in range(3 + step)]
        assert seq[step] == want
    assert len(seq[3]) == 1400 and seq[3][0] == 30000 and seq[3][700] == 30100


def test_dedup_with_tolerance():
    r = _recs([(1, 11, 1000), (1, 11, 1030), (1, 11, 5000), (1, 12, 1001)])
    r["bytes"][:, 0] = [7, 7, 7, 7]
    r["len"] = 5
    assert len(sdist.dedup_records(r, tol=0)) == 4
    d = sdist.dedup_records(r, tol=64)
    assert [int(x["sample_index"]) for x in d] == [1000, 5000, 1001]
    r["bytes"][1, 1] = 9                                       # different payload: not a duplicate
    assert len(sdist.dedup_records(r, tol=64)) == 4


def test_dedup_records_matches_the_three_key_definition():
    """dedup_records (packed-key sort) == sort by (proto, channel, sample_index) + adjacent-duplicate
    rule, with and without the 802.15.4 tolerance, also when sample_index does not fit the packed key."""
    from snout_amd._ffi import PKT_DTYPE
    from snout_amd.dist import dedup_records
    rng = np.random.default_rng(3)

    def reference(rec, tol):
        order = np.lexsort((rec["sample_index"], rec["channel"], rec["proto"]))
        r = rec[order]
        keep = [0]
        for i in range(1, r.size):
            j = i - 1                       # the record just before it in sorted order
            dup = (r["proto"][i] == r["proto"][j] and r["channel"][i] == r["channel"][j]
                   and int(r["sample_index"][i]) - int(r["sample_index"][j]) <= tol)
            if dup and tol > 0:
                dup = r["len"][i] == r["len"][j] and bytes(r["bytes"][i]) == bytes(r["bytes"][j])
            if not dup:
                keep.append(i)
        return r[keep]

    for base in (0, (1 << 52)):
        n = 3000
        rec = np.zeros(n, dtype=PKT_DTYPE)
        rec["proto"] = rng.integers(0, 2, n)
        rec["channel"] = rng.integers(0, 40, n)
        rec["sample_index"] = base + rng.integers(0, 200000, n).astype(np.uint64)
        rec["len"] = rng.integers(5, 40, n)
        rec["bytes"][:, :8] = rng.integers(0, 256, (n, 8))
        dup = rec[:600].copy()
        dup["sample_index"][300:] += rng.integers(1, 700, 300).astype(np.uint64)      # near-duplicates
        allrec = np.concatenate([rec, dup])
        rng.shuffle(allrec)
        for tol in (0, 520):
            assert np.array_equal(dedup_records(allrec.copy(), tol=tol), reference(allrec, tol)), (base, tol)
    assert dedup_records(np.zeros(0, dtype=PKT_DTYPE)).size == 0
