"""GPU: bench.py's contract — one JSON line with the required keys at N = 1, and the multi-rank
orchestration (rank env, per-step record gather, barrier, max over ranks) with two ranks.  The box
has one GPU, so the two ranks share it and talk over gloo (SNOUT_BENCH_BACKEND): a check of the
code path, not a measurement; the nccl pieces are covered at world size 1 in test_btle_gpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out: bytes) -> dict:
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--samples", "6e7", "--cpu-samples", "2e7"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert rf["traffic"] is None                       # the PMC figure is for the 1e9-sample workload only
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and d["value"] > 20 * cb["value"]


def test_two_ranks_share_the_gpu_over_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SNOUT_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--samples", "6e7"], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" not in d      # the CPU baseline is timed at N = 1 only
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["samples_per_gpu"] == 60000000


@pytest.mark.parametrize("workload,samples,cpu,name", [
    ("cfg3", 40 * (1 << 18), 40 * (1 << 16), "pfb_channelize<40>"),
    ("cfg4", 16 * (1 << 19), 16 * (1 << 17), "pfb_channelize<16>"),
    ("zigbee1", 1 << 23, 1 << 21, "zb_discrim..zb_walk"),
])
def test_other_workloads_keep_the_contract(workload, samples, cpu, name):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3",
                        "--warmup", "1", "--samples", str(samples), "--cpu-samples", str(cpu)],
                       capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["config"]["workload"].startswith(workload.replace("zigbee1", "single-channel 802"))
    assert d["roofline"]["kernel"] == name and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["traffic"] is None
    c = d["config"]
    assert c["decoded_crc_ok_per_gpu"] >= c["expected_crc_ok_per_gpu"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["value"] > d["cpu_baseline"]["value"]


def test_integer_input_format_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--format", "sc8", "--steps", "3",
                        "--warmup", "1", "--samples", "6e7", "--cpu-samples", "2e7"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["dtype"] == "i8->i32" and "sc8" in d["config"]["workload"]
    # 2 B per sample: the algorithmic bytes of the roofline follow the format
    assert abs(d["roofline"]["algorithmic_bytes"] - (2 * 6e7 + 160 * d["config"]["packets_per_gpu"])) < 1
    assert d["config"]["decoded_crc_ok_per_gpu"] >= d["config"]["expected_crc_ok_per_gpu"] > 0
