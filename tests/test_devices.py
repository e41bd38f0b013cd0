"""CPU: device table, `.b` dump and 802.15.4 MHR parse."""
import numpy as np
import pytest

from snout_amd import devices, formats
from snout_amd.message import BtleMessage


def _line(ts, n, mac, data_hex, crc="CRC0"):
    return (f"{ts:.6f} Pkt{n} Ch37 AA:8e89bed6 ADV_PDU_t0:ADV_IND T1 R0 PloadL{6 + len(data_hex) // 2} "
            f"AdvA:{mac} Data:{data_hex} {crc}\n").encode()


APPLE_NEARBY_LOCKED = "02011a0aff4c001005031c0b4c89"      # flags, Apple: Nearby, action 3, iOS 12 hint
APPLE_NEARBY_ACTIVE = "02011a0aff4c0010050b1c0b4c89"      # action 11
MS = "06ff0600010920"                                      # Microsoft


def test_device_table_rows_and_fingerprints(tmp_path):
    lines = [
        _line(1000.0, 0, "aabbccddeeff", APPLE_NEARBY_LOCKED),
        _line(1001.0, 1, "112233445566", MS),
        _line(1002.5, 2, "aabbccddeeff", APPLE_NEARBY_LOCKED),
        _line(1070.0, 3, "aabbccddeeff", APPLE_NEARBY_ACTIVE),
        _line(1071.0, 4, "deadbeef0001", "0201060303aafe", crc="CRC1"),     # rejected like the reference
    ]
    p = tmp_path / "scan.b"
    with open(p, "wb") as f:
        for ln in lines:
            if BtleMessage.fromraw(ln):
                f.write(ln)
    msgs = devices.read_b_dump(str(p))
    assert [m.number for m in msgs] == [0, 1, 2, 3]
    rows = devices.DeviceTable().extend(msgs).rows(now=1080.0)
    assert [r[0] for r in rows] == ["aabbccddeeff", "112233445566"]        # most recent first
    apple, ms = rows
    assert apple[2] == 3 and apple[3] == "01:10" and apple[4] == "Apple, Inc." and apple[6] == "iOS 12"
    assert apple[7] == "1 minute ago: Locked Screen, 10 seconds ago: Active User"
    assert apple[1] == "10 seconds ago"
    assert ms[2] == 1 and ms[3] == "00:00" and ms[4] == "Microsoft" and ms[6].startswith("Windows 10")
    assert ms[5] == "-" and ms[7] == "-"
    text = devices.DeviceTable().extend(msgs).render(now=1080.0)
    assert all(c in text for c in devices.COLUMNS) and "aabbccddeeff" in text


def test_uptime_formats():
    d = devices.DeviceEntry("x")
    assert d.uptime == -1 and d.uptime_nice == "-"
    for ts in (0.0, 3725.4):
        d.add(BtleMessage.fromraw(_line(ts, 0, "x" * 12, "020106")))
    assert d.uptime == 3725 and d.uptime_nice == "01:02:05"


def test_scan_writes_b_dump_that_reads_back(tmp_path):
    """BtleScan's save file (btle.py:105-106) holds exactly the accepted lines."""
    from snout_amd.scan import BtleScan
    path = tmp_path / "x.b"
    sc = BtleScan(filename=str(path), timeout=None)
    ok = [_line(5.0 + i, i, "0a0b0c0d0e0f", "020106") for i in range(3)]
    for ln in ok + [_line(9.0, 9, "0a0b0c0d0e0f", "020106", crc="CRC1")]:
        sc.handle_packet(ln)
    sc.conclude()
    assert open(path, "rb").read() == b"".join(ok)
    assert [m.timestamp for m in devices.read_b_dump(str(path))] == [5.0, 6.0, 7.0]


def test_parse_mhr_matches_the_frames_the_synthesizer_builds():
    from snout_amd import synth
    rng = np.random.default_rng(1)
    # data frame, PAN compression, short addresses: fc 0x8841, seq, dst pan, dst, src
    mpdu = bytes([0x41, 0x88, 0x2A, 0x34, 0x12, 0xFF, 0xFF, 0x01, 0x00]) + b"hello"
    mpdu += synth.crc16_154(mpdu).to_bytes(2, "little")
    h = formats.parse_mhr(mpdu)
    assert h["frame_type_name"] == "Data" and h["seq"] == 0x2A and h["panid_compress"]
    assert (h["dest_pan"], h["dest_addr"], h["src_pan"], h["src_addr"]) == (0x1234, 0xFFFF, 0x1234, 0x0001)
    assert mpdu[h["header_len"]:-2] == b"hello"
    # beacon request command without source address; extended source without compression
    h = formats.parse_mhr(bytes([0x03, 0x08, 0x07, 0xFF, 0xFF, 0xFF, 0xFF, 0x07, 0x00, 0x00]))
    assert h["frame_type_name"] == "Command" and h["src_addr"] is None and h["dest_addr"] == 0xFFFF
    ext = bytes([0x01, 0xC8, 0x01, 0x22, 0x11, 0x02, 0x00, 0x44, 0x33]) + bytes(range(8)) + b"\0\0"
    h = formats.parse_mhr(ext)
    assert h["src_mode"] == 3 and h["src_pan"] == 0x3344 and h["src_addr"] == int.from_bytes(bytes(range(8)), "little")
    ack = bytes([0x02, 0x00, 0x55, 0x00, 0x00])
    assert formats.parse_mhr(ack)["header_len"] == 3 and formats.parse_mhr(ack)["frame_type_name"] == "Ack"
    with pytest.raises(ValueError):
        formats.parse_mhr(bytes([0x41, 0x88, 0x01, 0x34]))
