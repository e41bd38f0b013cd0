"""Consumer contracts of the path, pinned by what the REFERENCE's own parser does
(tests/golden/btle_lines.json was produced by importing snout/core/message.py from
/root/reference — see tests/golden/make_golden.py)."""
import json
import os
import struct

import numpy as np

from snout_amd.message import BtleMessage
from snout_amd.rx import btle_format_line, rftap_encap
from snout_amd._ffi import PKT_DTYPE

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_fromraw_matches_reference_parser():
    cases = json.load(open(os.path.join(GOLD, "btle_lines.json")))
    assert sum(c["accepted"] for c in cases) >= 9
    for c in cases:
        m = BtleMessage.fromraw(c["line"].encode())
        assert bool(m) == c["accepted"], c["line"]
        if not m:
            continue
        assert m.sender == c["sender"]
        assert m.pdu_type == c["pdu_type"]
        assert m.timestamp == c["timestamp"]
        assert m.number == c["number"]
        assert m.payload_hex == c["payload"]["hex"]


def test_reference_docstring_example():
    # snout/core/message.py:214 — the one hot-path fixture the reference holds
    line = (b"1567108496.651985 Pkt8 Ch37 AA:8e89bed6 ADV_PDU_t0:ADV_IND T1 R0 PloadL20 "
            b"AdvA:6385725ebfcd Data:0201060aff4c001005011c569415 CRC1\n")
    assert BtleMessage.fromraw(line) is False
    m = BtleMessage.fromraw(line.replace(b"CRC1", b"CRC0"))
    assert (m.sender, m.channel, m.pdu_type, m.number) == ("6385725ebfcd", "37", "ADV_IND", 8)


def test_formatter_reproduces_docstring_line():
    """Build the packet of message.py:214 by hand and format it: byte-identical line."""
    p = np.zeros(1, dtype=PKT_DTYPE)[0]
    adva = bytes.fromhex("6385725ebfcd")[::-1]
    data = bytes.fromhex("0201060aff4c001005011c569415")
    body = bytes([0x40, 6 + len(data)]) + adva + data
    p["proto"], p["channel"], p["len"] = 0, 37, len(body) + 3
    p["pdu_type"], p["flags"], p["crc_ok"] = 0, 1, 0
    p["bytes"][:len(body)] = np.frombuffer(body, dtype=np.uint8)
    p["sample_index"] = int(round(0.651985 * 4e6))
    line = btle_format_line(p, 4e6, 1567108496.0, 8)
    assert line == (b"1567108496.651985 Pkt8 Ch37 AA:8e89bed6 ADV_PDU_t0:ADV_IND T1 R0 PloadL20 "
                    b"AdvA:6385725ebfcd Data:0201060aff4c001005011c569415 CRC1\n")


def test_formatter_regenerates_golden_lines(oracle):
    cases = json.load(open(os.path.join(GOLD, "btle_lines.json")))
    x = np.fromfile(os.path.join(GOLD, "btle_ch37_4msps.cf32"), dtype=np.complex64)
    pk, _ = oracle.btle_segment(x, channel=37)
    ours = [btle_format_line(p, 4e6, 1567108496.0, i).decode() for i, p in enumerate(pk)]
    assert ours == [c["line"] for c in cases[5:13]]
    for ln in ours:
        assert len(ln.split(" ")) == 11 and ln.endswith(" CRC0\n")


def test_other_pdu_types_keep_the_grammar():
    p = np.zeros(1, dtype=PKT_DTYPE)[0]
    p["proto"], p["channel"], p["len"], p["crc_ok"] = 0, 38, 2 + 12 + 3, 1
    p["bytes"][:14] = np.arange(14, dtype=np.uint8)
    for t, first in [(1, "A0:"), (3, "ScanA:"), (4, "AdvA:"), (6, "AdvA:")]:
        p["pdu_type"] = t
        tok = btle_format_line(p, 4e6, 0.0, 1).decode().split(" ")
        assert len(tok) == 11 and tok[8].startswith(first) and tok[2] == "Ch38"
    p["pdu_type"] = 5   # CONNECT_REQ: more tokens -> the reference parser drops it
    assert len(btle_format_line(p, 4e6, 0.0, 1).decode().split(" ")) != 11


def test_rftap_datagram_layout():
    """rftap_encap(2,195,'') with meta {qual}: 'RFta', len32=4, flags=DLT|QUAL, dlt=195, f32 qual,
    then the MPDU (top_block.py:53; epy_block_0.py:20-24 qual = lqi/255)."""
    p = np.zeros(1, dtype=PKT_DTYPE)[0]
    mpdu = bytes.fromhex("03083affffffff07aabb")
    p["proto"], p["channel"], p["len"], p["lqi"] = 1, 11, len(mpdu), 204
    p["bytes"][:len(mpdu)] = np.frombuffer(mpdu, dtype=np.uint8)
    d = rftap_encap(p)
    assert d[:4] == b"RFta"
    len32, flags, dlt = struct.unpack("<HHI", d[4:12])
    assert (len32, flags, dlt) == (4, 0x0081, 195)
    assert struct.unpack("<f", d[12:16])[0] == np.float32(204) / np.float32(255)
    assert d[16:] == mpdu and len(d) == 4 * len32 + len(mpdu)


def test_integer_sample_formats_are_exact_in_float32():
    """include/snout_rx.h: an sc8 / sc16 sample v stands for v * 2^-7 / v * 2^-15.  Both the conversion
    and the scale are exact in float32, so the integer paths can be checked bit for bit against the
    cf32 oracle on the converted capture."""
    import numpy as np
    from oracle import oracle_py
    from snout_amd import synth
    for dt, bits in ((np.int8, 7), (np.int16, 15)):
        info = np.iinfo(dt)
        v = np.arange(info.min, info.max + 1, dtype=dt)
        f = oracle_py.from_int(v)
        assert f.dtype == np.float32
        assert np.array_equal(f.astype(np.float64) * 2.0 ** bits, v.astype(np.float64))
    x = (np.linspace(-1, 1, 4001) + 1j * np.linspace(1, -1, 4001)).astype(np.complex64)
    for fmt, dt in ((1, np.int8), (2, np.int16)):
        q = synth.quantize(x, fmt)
        assert q.dtype == dt and q.size == 2 * x.size
        assert q.max() <= np.iinfo(dt).max * 0.81 and q.min() >= np.iinfo(dt).min * 0.81   # 1.25x headroom
        err = np.abs(oracle_py.from_int(q) * 1.25 - x.view(np.float32))
        assert err.max() <= 1.25 * 2.0 ** -(7 if fmt == 1 else 15) * 0.5 + 1e-7
