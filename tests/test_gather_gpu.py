"""GPU: rank 0's side of the record gather in the library -- `snout_records_dedup` (sort + duplicate removal on the device)
against the host statement of the rule (`snout_amd.dist.dedup_records`), and the exchange object driving it."""
import ctypes as C

import numpy as np
import pytest

from snout_amd import dist as sdist

pytestmark = pytest.mark.gpu


def _blocks(rng, width, tol, world, cap, counts):
    dt = sdist.wire_dtype(width)
    blocks = np.zeros((world, cap), dtype=dt)
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
        if r == world - 1 and counts[0] >= 200 and n >= 200:    # near-duplicates of rank 0's records
            b[:200] = blocks[0][:200]
            b["sample_index"][:200] += rng.integers(0, 3, 200).astype(np.uint64) * (260 if tol else 0)
        full.append(b[:n])
    if counts[0] > 5:
        blocks[0]["sample_index"][5] = int(sdist._DROP)     # a record its segment disowned
    want_in = np.concatenate(full)
    want_in = want_in[want_in["sample_index"] < int(sdist._DROP)]
    return dt, blocks, sdist.dedup_records(sdist.widen_records(want_in), tol=tol)


@pytest.mark.parametrize("width,tol,world,cap,counts", [
    (80, 0, 3, 700, [650, 0, 333]), (160, 520, 3, 700, [650, 0, 333]), (96, 0, 8, 5000, [5000, 4999, 1, 0, 3000, 5000, 2500, 4000]),
    (160, 520, 1, 300, [300]), (32, 0, 2, 64, [0, 0]), (160, 520, 8, 1200, [1200] * 8)])
def test_library_dedup_equals_host_dedup(width, tol, world, cap, counts):
    import torch
    from snout_amd import _ffi
    lib = _ffi.load()
    rng = np.random.default_rng(9 + world)
    dt, blocks, want = _blocks(rng, width, tol, world, cap, counts)
    rows = torch.from_numpy(blocks.view(np.uint8).reshape(-1).copy()).cuda()
    # counts as the exchange holds them: one 4 x int64 header per rank; a count above the capacity is clamped
    heads = torch.zeros((world, 4), dtype=torch.int64)
    heads[:, 0] = torch.tensor(counts)
    if counts[0] == cap:
        heads[0, 0] = cap + 77
    heads = heads.cuda()
    work = torch.empty(lib.snout_records_dedup_workspace(world, cap), dtype=torch.uint8, device="cuda")
    out = torch.zeros(world * cap * width, dtype=torch.uint8, device="cuda")
    n_keep = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):                                  # the workspace is reusable
        _ffi.check(lib.snout_records_dedup(C.c_void_p(rows.data_ptr()), width, world, cap, C.c_void_p(heads.data_ptr()), 4, tol,
                                           C.c_void_p(out.data_ptr()), C.c_void_p(n_keep.data_ptr()), C.c_void_p(work.data_ptr()),
                                           work.numel(), C.c_void_p(st)))
    torch.cuda.synchronize()
    n = int(n_keep.item())
    got = sdist.widen_records(out.cpu().numpy()[:n * width].view(dt))
    assert n == len(want)
    for f in ("proto", "channel", "sample_index", "len", "crc_ok", "lqi", "flags", "aux"):
        assert np.array_equal(got[f], want[f]), (width, f)
    assert np.array_equal(got["bytes"], want["bytes"])
    # too small a workspace and a width that is not a multiple of 16 are refused
    assert lib.snout_records_dedup(C.c_void_p(rows.data_ptr()), width, world, cap, C.c_void_p(heads.data_ptr()), 4, tol,
                                   C.c_void_p(out.data_ptr()), C.c_void_p(n_keep.data_ptr()), C.c_void_p(work.data_ptr()),
                                   16, C.c_void_p(st)) == -1
    assert lib.snout_records_dedup(C.c_void_p(rows.data_ptr()), 40, world, cap, C.c_void_p(heads.data_ptr()), 4, tol,
                                   C.c_void_p(out.data_ptr()), C.c_void_p(n_keep.data_ptr()), C.c_void_p(work.data_ptr()),
                                   work.numel(), C.c_void_p(st)) == -1


@pytest.mark.parametrize("tol,fake", [(0, 0), (520, 0), (0, 4), (None, 3)])
def test_exchange_object_on_the_gpu(tol, fake):
    """AsyncRecordGather without a process group on a GPU (device copy instead of the collective): host records in, the
    library's dedup behind them, pinned host records out; `fake_world` multiplies rank 0's blocks."""
    import torch
    rng = np.random.default_rng(3)
    dt, blocks, want = _blocks(rng, 160, tol or 0, 1, 900, [900])
    recs = sdist.widen_records(blocks[0])
    g = sdist.AsyncRecordGather(torch.device("cuda", 0), width=160, dedup_tol=tol, cap=1000, fake_world=fake)
    outs = []
    for step in range(5):
        if len(g.inflight) == 2:
            outs.append(g.finish())
        g.begin()
        g.append(recs[:400])
        g.append(recs[400:])
        g.launch()
    while g.inflight:
        outs.append(g.finish())
    assert len(outs) == 5
    for o in outs:
        o = sdist.widen_records(o)
        if tol is None:
            assert len(o) == 900 * max(1, fake)
            continue
        assert len(o) == len(want) * max(1, fake)
        first = o[o["sample_index"] < (1 << 40)]
        order = np.lexsort((first["sample_index"], first["channel"], first["proto"]))
        assert np.array_equal(order, np.arange(len(first)))
        assert first.tobytes() == want.tobytes()


@pytest.mark.parametrize("tol", [None, 0])
def test_downloads_sized_by_the_exchange_before_still_deliver_everything(tol):
    """Rank 0 downloads what the exchange before delivered (+ 2 %): when an exchange brings more, finish() fetches the
    rest -- nothing is ever cut off."""
    import torch
    rng = np.random.default_rng(4)
    dt, blocks, _ = _blocks(rng, 160, 0, 1, 3000, [3000])
    recs = sdist.widen_records(blocks[0])
    recs["sample_index"] = np.arange(len(recs), dtype=np.uint64) * 7           # all distinct: nothing de-duplicates away
    recs["channel"] = 3
    recs["proto"] = 1                                                            # one sort key order = the order appended
    g = sdist.AsyncRecordGather(torch.device("cuda", 0), width=160, dedup_tol=tol, cap=3200, fake_world=2)
    sizes = [100, 100, 2900, 2900, 40, 3000, 0, 1500]
    outs = []
    for n in sizes:
        if len(g.inflight) == 2:
            outs.append(g.finish())
        g.begin()
        g.append(recs[:n])
        g.launch()
    while g.inflight:
        outs.append(g.finish())
    assert [len(o) for o in outs] == [2 * n for n in sizes]
    for n, o in zip(sizes, outs):
        o = sdist.widen_records(o)
        first = o[o["sample_index"] < (1 << 40)]
        assert first.tobytes() == recs[:n].tobytes()


def test_a_capture_without_segments_for_this_rank_takes_its_turn():
    """a rank that has no segment of a capture used to call on_first / on_last immediately inside start(), i.e. while
    the exchange of the capture BEFORE was still being filled (AsyncRecordGather.begin() then asserts).  The callbacks now ride
    on an empty job and fire in the collection order.  And a scan that keeps its records on the device refuses to be collected
    to the host."""
    import torch
    from snout_amd import synth
    from snout_amd.sharded import ShardedScan
    x, _ = synth.wideband_capture(0, 40 * (1 << 16), seed=5, sigma=0.02)
    cap = torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).cuda()
    n_in = cap.numel() // 2
    src = lambda a, b: cap[2 * a:2 * b]
    sc = ShardedScan(0, n_channels=40, seg_len=40 * (1 << 14), batch=2)
    events = []
    sc.start(n_in, src, on_first=lambda: events.append("A first"), on_last=lambda: events.append("A last"))
    mine = sc.my_segments
    sc.my_segments = lambda n_total, group=None: []          # the next capture has nothing for this rank
    sc.start(n_in, src, on_first=lambda: events.append("B first"), on_last=lambda: events.append("B last"))
    sc.my_segments = mine
    sc.start(n_in, src, on_first=lambda: events.append("C first"), on_last=lambda: events.append("C last"))
    assert events == []                                      # nothing fires inside start()
    while sc.active():
        sc.step()
    assert events == ["A first", "A last", "B first", "B last", "C first", "C last"]
    sc.close()
    dev = ShardedScan(0, n_channels=40, seg_len=40 * (1 << 14), batch=2, records_on_device=True)
    with pytest.raises(RuntimeError, match="records on the device"):
        dev.run(n_in, src)
    dev.close()
