"""CPU: the parts of bench.py that run without a GPU — the launcher's GPU count from the KFD topology (the parent of
`bench.py --gpus N` never opens the HIP runtime), and the synthetic captures' tile schedule."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _topology(tmp_path, simd_counts):
    for i, simd in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ngfx_target_version 90500\n")
    return str(tmp_path)


def test_visible_gpus_counts_kfd_nodes_with_simds(tmp_path, monkeypatch):
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    sysfs = _topology(tmp_path, [0, 0, 1024, 1024, 1024, 1024])     # two CPU nodes, four GPUs
    assert bench.visible_gpus(sysfs) == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")                    # exported empty (as in the build container): not a list
    assert bench.visible_gpus(sysfs) == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus(sysfs) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3,7")          # an index beyond the topology does not count
    assert bench.visible_gpus(sysfs) == 4
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-deadbeef")       # a UUID entry counts as one device
    assert bench.visible_gpus(sysfs) == 1


def test_visible_gpus_is_unknown_without_a_topology(tmp_path):
    assert bench.visible_gpus(str(tmp_path / "absent")) is None


def test_self_launch_refuses_more_ranks_than_gpus(tmp_path, monkeypatch, capsys):
    sysfs = _topology(tmp_path, [0, 1024])
    count = bench.visible_gpus
    monkeypatch.setattr(bench, "visible_gpus", lambda: count(sysfs))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.delenv("SNOUT_BENCH_BACKEND", raising=False)
    assert bench.self_launch(8) == 2                                 # fails fast, before any child is started
    assert "1 GPU(s) visible" in capsys.readouterr().err


def test_wideband_tiles_composed_in_torch_equal_the_numpy_compositor():
    """bench.wideband_tile_dev (the device compositor of the benchmark's captures) against snout_amd.synth.wideband_capture:
    the same narrowband streams and truth lists, samples equal to ~1e-6 of full scale (f32 instead of complex128 filtering)."""
    import numpy as np
    import torch
    from snout_amd import synth
    for proto, n in ((0, 40 * 4096), (1, 16 * (1 << 14))):
        a, ta = synth.wideband_capture(proto, n, seed=305, sigma=0.0)
        b, tb = bench.wideband_tile_dev(proto, n, 305, torch.device("cpu"))
        assert [(t.channel, t.sample_index, t.payload) for t in ta] == [(t.channel, t.sample_index, t.payload) for t in tb]
        assert np.abs(a - b.numpy()).max() < 1e-5 * max(1.0, float(np.abs(a).max()))


def test_a_capture_cycles_through_distinct_tiles():
    """the virtual capture's tile g is tile g mod K of K independently seeded tiles; a window that
    starts anywhere reads the right stretch of each, and the truth count follows the whole tiles inside it."""
    import numpy as np
    import torch
    K, L = 4, 64
    tiles = torch.arange(K * 2 * L, dtype=torch.float32).reshape(K, 2 * L)
    gen = torch.Generator()
    gen.manual_seed(1)
    sigma, bench.SIGMA = bench.SIGMA, 0.0
    try:
        x = torch.empty(2 * (5 * L + 10))
        bench.fill_capture(x, 3 * L - 7, tiles, gen)
    finally:
        bench.SIGMA = sigma
    virt = torch.cat([tiles[g % K] for g in range(10)])
    assert torch.equal(x, virt[2 * (3 * L - 7):2 * (3 * L - 7) + x.numel()])
    truths = [[0] * (g + 1) for g in range(K)]             # tile g holds g + 1 packets
    assert bench.truth_in(truths, L, 5 * L) == 1 + 2 + 3 + 4 + 1
    assert bench.truth_in(truths, L, 2 * L + 5, first_sample=L - 3) == 2 + 3    # whole tiles 1 and 2
