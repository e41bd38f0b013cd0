"""CPU: the hand-scheduled LDS reads of the channelizer (pfb_spec.hip issues `ds_read_b64` in one asm statement and waits in
a later one) must never be touched by the compiler in between -- checked on the ISA of the shipped build.  hipcc
cross-compiles gfx950 without a GPU."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_lds_asm", os.path.join(ROOT, "tools", "check_lds_asm.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_the_checker_sees_a_premature_use():
    ok = """
_Zk:
	ds_read_b64 v[2:3], v1 offset:8
	ds_read_b64 v[4:5], v1 offset:16
	v_add_f32_e32 v9, v8, v7
	s_waitcnt lgkmcnt(1)
	v_add_f32_e32 v9, v2, v3
	s_waitcnt lgkmcnt(0)
	v_pk_fma_f32 v[10:11], v[4:5], v[2:3], v[10:11]
	s_endpgm
"""
    assert chk.check(ok) == ([], 2)
    early_read = ok.replace("v_add_f32_e32 v9, v8, v7", "v_mov_b32_e32 v9, v5")
    early_write = ok.replace("v_add_f32_e32 v9, v8, v7", "v_mov_b32_e32 v3, v8")
    partial = ok.replace("v_add_f32_e32 v9, v2, v3", "v_add_f32_e32 v9, v4, v3")       # lgkmcnt(1): only the OLDER read is back
    for bad in (early_read, early_write, partial):
        problems, _ = chk.check(bad)
        assert len(problems) == 1
    # LDS writes count in lgkmcnt as well: behind read, write, lgkmcnt(1) the read is back
    wr = ok.replace("v_add_f32_e32 v9, v8, v7\n\ts_waitcnt lgkmcnt(1)\n\tv_add_f32_e32 v9, v2, v3",
                    "ds_write_b64 v1, v[20:21]\n\ts_waitcnt lgkmcnt(1)\n\tv_add_f32_e32 v9, v2, v5")
    assert wr != ok and chk.check(wr)[0] == []
    # a scalar load in flight shares the counter and returns out of order: only lgkmcnt(0) counts then
    smem = ok.replace("ds_read_b64 v[4:5], v1 offset:16", "ds_read_b64 v[4:5], v1 offset:16\n\ts_load_dwordx2 s[0:1], s[4:5], 0x0")
    assert len(chk.check(smem)[0]) == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_use_of_an_lds_read_before_its_wait_in_the_shipped_channelizer(tmp_path):
    src = os.path.join(ROOT, "snout_amd", "csrc", "pfb_spec.hip")
    out = str(tmp_path / "pfb_spec.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                        "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", out, src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    assert text.count("ds_read_b64") > 300                     # the hand-written reads are in there
    problems, n_reads = chk.check(text)
    assert n_reads > 1000 and problems == [], problems[:5]
