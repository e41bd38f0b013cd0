"""GPU: guards against the two regressions round 5 shipped unnoticed (profiles/r6_streams.md) -- measured
relations with generous margins, not absolute times:
  * rank 0's load at N = 8, rehearsed on this one GPU (SNOUT_BENCH_FAKE_WORLD=8 on the RCCL backend at world 1), must stay
    under the next step: round 4 + 9 %, round 5 + 41 %, round 6 + 5 %;
  * `btle_corr_planes` sits between two channelizer launches on the front stream: the step may exceed the channelizer's
    event duration by its ~0.08 ms and the launch gaps, not by round 5's 0.26 ms."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(env_more, *args):
    env = dict(os.environ, **env_more)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-others", "--steps", "20", "--warmup", "3", *args],
                       capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])


def test_rank0s_exchange_of_eight_blocks_stays_under_the_next_step():
    one = _bench({"SNOUT_BENCH_NCCL1": "1", "SNOUT_BENCH_FAKE_WORLD": "0"})
    eight = _bench({"SNOUT_BENCH_NCCL1": "1", "SNOUT_BENCH_FAKE_WORLD": "8"})
    assert eight["config"]["records_on_rank0_last_step"] == 8 * one["config"]["records_on_rank0_last_step"] > 400000
    assert eight["ms_per_step"] <= 1.15 * one["ms_per_step"], (one["ms_per_step"], eight["ms_per_step"])


def test_the_headline_step_is_the_channelizer_plus_the_correlator():
    d = _bench({})
    gap = d["ms_per_step"] - d["roofline"]["kernel_ms"]
    assert 0.0 < gap <= 0.20, (d["ms_per_step"], d["roofline"]["kernel_ms"])
