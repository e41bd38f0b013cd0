"""GPU: the C ABI used from plain C (gcc, no Python in the data path): tests/c/btle_rx_c.c links
libsnout_rx.so through include/snout_rx.h alone, decodes the golden capture and prints the same
btle_rx lines the ctypes path produces."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "snout_amd", "lib")


def test_plain_c_program_decodes_the_golden_capture(tmp_path):
    exe = str(tmp_path / "btle_rx_c")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "btle_rx_c.c"), "-o", exe,
                           "-L", LIBDIR, "-lsnout_rx", "-Wl,-rpath," + LIBDIR])
    cap = os.path.join(ROOT, "tests", "golden", "btle_ch37_4msps.cf32")
    t0 = 1567108496.0
    r = subprocess.run([exe, cap, "37", repr(t0)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.splitlines(keepends=True)
    truth = json.load(open(os.path.join(ROOT, "tests", "golden", "btle_ch37_truth.json")))
    assert len(lines) == len(truth) == 8
    # the same lines through the Python binding
    from snout_amd.rx import SnoutRx, btle_format_line
    x = np.fromfile(cap, dtype=np.complex64)
    with SnoutRx(proto=0, channel=37) as rx:
        want = [btle_format_line(p, 4e6, t0, i, 0x8E89BED6) for i, p in enumerate(rx.process(x))]
    assert lines == want
    from snout_amd.message import BtleMessage
    for ln, t in zip(lines, truth):
        m = BtleMessage.fromraw(ln)
        assert m and ln.endswith(b"CRC0\n") and f"Ch{t['channel']}".encode() in ln
        assert abs(m.timestamp - (t0 + t["sample_index"] / 4e6)) < 1e-3


def test_plain_c_program_on_hackrf_int8_iq(tmp_path):
    """The same C program fed interleaved int8 IQ (SNOUT_FMT_SC8), the format upstream btle_rx reads."""
    from snout_amd import synth
    exe = str(tmp_path / "btle_rx_c")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "btle_rx_c.c"), "-o", exe,
                           "-L", LIBDIR, "-lsnout_rx", "-Wl,-rpath," + LIBDIR])
    x = np.fromfile(os.path.join(ROOT, "tests", "golden", "btle_ch37_4msps.cf32"), dtype=np.complex64)
    cap = tmp_path / "btle_ch37.sc8"
    synth.quantize(x, 1).tofile(cap)
    r = subprocess.run([exe, str(cap), "37", "1567108496.0", "sc8"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.splitlines(keepends=True)
    truth = json.load(open(os.path.join(ROOT, "tests", "golden", "btle_ch37_truth.json")))
    assert len(lines) == len(truth) == 8 and all(ln.endswith(b"CRC0\n") for ln in lines)
    from snout_amd.message import BtleMessage
    for ln, t in zip(lines, truth):
        pdu = bytes.fromhex(t["pdu"])
        m = BtleMessage.fromraw(ln)
        assert m.sender == pdu[2:8][::-1].hex() and m.payload_hex == pdu[8:].hex()
