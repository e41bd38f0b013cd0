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


def test_group_submissions_batches_equal_lengths_in_order():
    """What one batched submission may carry: consecutive segments of equal length, at most `batch`."""
    n, seg, ov, pre = 10_000, 1024, 100, 256
    for world in (1, 2, 3):
        for rank in range(world):
            segs = sdist.shard_segments(n, seg, ov, rank, world, pre)
            for batch in (1, 2, 4, 8):
                runs = sdist.group_submissions(segs, batch)
                assert [j for run in runs for j in run] == list(range(len(segs)))          # every segment once, in order
                assert all(1 <= len(run) <= batch for run in runs)
                for run in runs:
                    assert len({segs[j][1] - segs[j][0] for j in run}) == 1                  # equal lengths inside a run
                for a, b in zip(runs, runs[1:]):                                             # runs are maximal
                    same = segs[b[0]][1] - segs[b[0]][0] == segs[a[0]][1] - segs[a[0]][0]
                    assert not same or len(a) == batch
    assert sdist.group_submissions([], 4) == []
    assert sdist.group_submissions([(0, 10), (10, 20), (20, 25), (25, 35)], 4) == [[0, 1], [2], [3]]


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


def _worker_overflow(rank, world, port, q):
    """A rank that outgrows the agreed capacity must not leave the others in the collective: every
    rank learns the true counts from the gathered headers, and the remainder travels in a second
    all_gather that all ranks enter together (overflow is a collective decision)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for dedup in (None, 0):
            g = sdist.AsyncRecordGather(width=80, dedup_tol=dedup)
            outs = []
            sizes = [(3, 5), (4000, 7), (6, 3000), (5000, 5200)]       # (rank 0, rank 1) records per step
            for step, sz in enumerate(sizes):
                r = _recs([(0, 37 + rank, 100000 * step + i) for i in range(sz[rank])])
                if len(g.inflight) == 2:
                    outs.append(g.finish())
                g.start(r)
            while g.inflight:
                outs.append(g.finish())
            if rank == 0:
                q.put((dedup, [(len(o), sorted(set(o["channel"].tolist())),
                               int(o["sample_index"].min()), int(o["sample_index"].max())) for o in outs], g.cap_scheduled))
            else:
                assert all(o is None for o in outs)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_overflow_is_a_collective_decision_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_overflow, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180), q.get(timeout=180)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sizes = [(3, 5), (4000, 7), (6, 3000), (5000, 5200)]
    for dedup, outs, cap in res:
        assert len(outs) == 4
        for step, (n, chans, lo, hi) in enumerate(outs):
            assert n == sum(sizes[step]) and chans == [37, 38], (dedup, step, n)
            assert lo == 100000 * step and hi == 100000 * step + max(sizes[step]) - 1
        assert cap >= 5200                         # the capacity grows with the traffic, on every rank alike (from a later exchange on)


def _worker_overflow_straddle(rank, world, port, q):
    """rank 0 has already OPENED exchanges 1 and 2 when exchange 0 overflows and its finish() grows the capacity;
    rank 1 opens them afterwards.  The capacity of an exchange goes by its index, so both ranks size exchanges 1 .. AHEAD
    alike (the old capacity), and the grown one holds from the same exchange on both."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sdist.AsyncRecordGather(width=80, cap=64)
        n_ex = 9
        sizes = [(5000, 3)] + [(70 + k, 4000 if k == 3 else 10) for k in range(1, n_ex)]    # (rank 0, rank 1) per exchange
        recs = [_recs([(0, 37 + rank, 100000 * k + i) for i in range(sizes[k][rank])]) for k in range(n_ex)]
        outs, caps = [], []

        def begin(k):
            g.begin(len(recs[k])); caps.append(g.cur["cap"]); g.append(recs[k]); g.close()
        # (launches and finishes come in the same order on both ranks, as they must: only WHEN an exchange is opened differs)
        if rank == 0:       # opens 0, 1, 2 before anything is finished ...
            begin(0); begin(1); begin(2)
            g.launch()
            outs.append(g.finish())         # ... exchange 0 overflowed: the capacity grows, in force from exchange 0 + AHEAD + 1
            g.launch(); g.launch()
            nxt = 3
        else:               # ... rank 1 opens 1 and 2 only after it has finished 0
            begin(0); g.launch()
            outs.append(g.finish())
            begin(1); begin(2); g.launch(); g.launch()
            nxt = 3
        for k in range(nxt, n_ex):
            while len(g.inflight) >= 2:
                outs.append(g.finish())
            begin(k); g.launch()
        while g.inflight:
            outs.append(g.finish())
        q.put((rank, caps, [None if o is None else len(o) for o in outs]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_capacity_goes_by_the_exchange_index_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_overflow_straddle, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (caps, outs)) for r, caps, outs in (q.get(timeout=180), q.get(timeout=180)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    caps0, outs0 = res[0]
    caps1, outs1 = res[1]
    assert caps0 == caps1                                   # every exchange has the same capacity on both ranks
    A = sdist.AsyncRecordGather.AHEAD
    assert caps0[:A + 1] == [64] * (A + 1) and caps0[A + 1] >= 5000 and caps0[-1] >= 5000
    sizes = [(5000, 3)] + [(70 + k, 4000 if k == 3 else 10) for k in range(1, 9)]
    assert outs0 == [a + b for a, b in sizes] and all(o is None for o in outs1)       # nothing lost in the overflowing ones


def _worker_order(rank, world, port, q):
    """Two gathers on one process group launched in different orders on the two ranks: their 32-byte header collectives
    have equal sizes and would pair up crosswise without anyone noticing; the launch ticket in the header makes finish()
    raise on every rank."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = sdist.AsyncRecordGather(width=80, cap=64)
        b = sdist.AsyncRecordGather(width=80, cap=64)
        r = _recs([(0, 37 + rank, i) for i in range(5)])
        for g in (a, b):
            g.begin(5); g.append(r); g.close()
        order = (a, b) if rank == 0 else (b, a)
        for g in order:
            g.launch()
        err = 0
        for g in order:
            try:
                g.finish()
            except RuntimeError as e:
                err += "different orders" in str(e)
        q.put((rank, err))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gathers_launched_in_different_orders_are_caught_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_order, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((q.get(timeout=180), q.get(timeout=180)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == 2 and res[1] == 2


def test_device_dedup_equals_host_dedup():
    """dedup_device (fixed-shape torch ops, what rank 0 runs on its GPU behind the all_gather) keeps
    exactly the records dedup_records keeps, for the BTLE (exact) and 802.15.4 (tolerance) rules."""
    import torch
    rng = np.random.default_rng(9)
    for width, tol in ((80, 0), (160, 520)):
        world, cap = 3, 700
        dt = sdist.wire_dtype(width)
        blocks = np.zeros((world, cap), dtype=dt)
        counts = [650, 0, 333]
        full = []
        for r in range(world):
            n = counts[r]
            b = blocks[r]
            b["proto"][:n] = rng.integers(0, 2, n)
            b["channel"][:n] = rng.integers(0, 5, n)
            b["sample_index"][:n] = rng.integers(0, 40000, n)
            b["len"][:n] = rng.integers(5, 30, n)
            b["bytes"][:n, :6] = rng.integers(0, 4, (n, 6))
            b["sample_index"][n:] = 7                       # garbage behind the valid prefix is ignored
            if r == 2:                                      # near-duplicates of rank 0's records
                b[:200] = blocks[0][:200]
                b["sample_index"][:200] += rng.integers(0, 3, 200).astype(np.uint64) * (260 if tol else 0)
            full.append(b[:n])
        blocks[0]["sample_index"][5] = int(sdist._DROP)     # a record its segment disowned
        want_in = np.concatenate(full)
        want_in = want_in[want_in["sample_index"] < int(sdist._DROP)]
        want = sdist.dedup_records(sdist.widen_records(want_in), tol=tol)
        rows = torch.from_numpy(blocks.view(np.uint8).reshape(world * cap, width).copy()).view(torch.int64)
        out, n_keep = sdist.dedup_device(torch, rows, torch.tensor(counts, dtype=torch.int64), cap, tol)
        got = out.numpy().view(np.uint8).reshape(-1, width)[:int(n_keep)].copy().view(dt).reshape(-1)
        got = sdist.widen_records(got)
        assert len(got) == len(want)
        for f in ("proto", "channel", "sample_index", "len"):
            assert np.array_equal(got[f], want[f]), (width, f)
        assert np.array_equal(got["bytes"], want["bytes"])


def test_gather_rejects_records_wider_than_the_wire_format():
    """Too wide for the wire: not an exception in append() (other ranks would sit in the all_gather) but a
    collective one -- the length travels in the header slot and finish() raises."""
    g = sdist.AsyncRecordGather(width=80)
    r = _recs([(1, 11, 5)])
    r["len"] = 100
    g.begin(1)
    g.append(r)
    g.launch()
    with pytest.raises(ValueError):
        g.finish()


def _worker_wide(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sdist.AsyncRecordGather(width=80)
        r = _recs([(0, 3 + rank, 1000 + i) for i in range(5)])
        if rank == 1:
            r["len"][2] = 68            # a false access-address match on a data channel: 2 + 63 + 3 bytes
        g.start(r)
        try:
            g.finish()
            q.put((rank, "no error"))
        except ValueError as e:
            q.put((rank, "ValueError"))
        dist.barrier()                  # both ranks are still in step: neither hangs in a collective
    finally:
        dist.destroy_process_group()


def test_too_wide_record_on_one_rank_raises_on_every_rank_gloo_world2():
    """only rank 1 holds the oversized record; both ranks must raise, in finish()."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_wide, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == {0: "ValueError", 1: "ValueError"}


def test_the_gathers_registry_holds_its_groups():
    """launch tickets and creation indices are kept per process GROUP OBJECT, not per id(): a new group that
    happens to get a destroyed group's id starts from zero, and the default group's counters restart when
    torch.distributed has been re-initialised."""
    from snout_amd.dist import AsyncRecordGather

    class FakeDist:
        class group:
            WORLD = None

        @staticmethod
        def is_initialized():
            return FakeDist.group.WORLD is not None

    class G:
        pass

    saved = dict(AsyncRecordGather._groups)
    try:
        AsyncRecordGather._groups.clear()
        g1 = G()
        e1 = AsyncRecordGather._registry(g1, FakeDist)
        e1[1] += 2
        e1[2] += 5
        assert AsyncRecordGather._registry(g1, FakeDist) is e1 and e1[0] is g1          # the entry keeps the group alive
        g2 = G()
        AsyncRecordGather._groups[id(g2)] = [g1, 7, 7]                                   # a stale entry under g2's id
        e2 = AsyncRecordGather._registry(g2, FakeDist)
        assert e2[0] is g2 and e2[1:] == [0, 0]                                          # not the stale counters
        FakeDist.group.WORLD = object()
        d1 = AsyncRecordGather._registry(None, FakeDist)
        d1[2] += 3
        assert AsyncRecordGather._registry(None, FakeDist) is d1
        FakeDist.group.WORLD = object()                                                  # destroy_process_group + init again
        assert AsyncRecordGather._registry(None, FakeDist)[1:] == [0, 0]
    finally:
        AsyncRecordGather._groups.clear()
        AsyncRecordGather._groups.update(saved)
