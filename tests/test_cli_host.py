"""Host logic that needs no GPU: channel parsing, pcap writer, scan stop rules with a fake source."""
import numpy as np
import pytest

from snout_amd import cli, formats


def test_parse_channels_like_reference():
    assert cli.parse_channels("37", "btle") == [37]
    assert cli.parse_channels("37,38,39", "btle") == [37, 38, 39]
    assert cli.parse_channels("11:14", "zigbee") == [11, 12, 13, 14]
    assert cli.parse_channels("", "zigbee") == [11]
    with pytest.raises(Exception):
        cli.parse_channels("40", "btle")
    with pytest.raises(Exception):
        cli.parse_channels("10", "zigbee")


def test_pcap_roundtrip(tmp_path):
    p = str(tmp_path / "z.pcap")
    frames = [(1567108496.25, bytes.fromhex("03083affffffff07aabb")), (1567108497.999999, b"\x01\x02\x03")]
    assert formats.write_pcap(p, frames) == 2
    assert formats.write_pcap(p, [(1567108498.5, b"\x09")], append=True) == 1
    lt, got = formats.read_pcap(p)
    assert lt == 195 and [g[1] for g in got] == [f[1] for f in frames] + [b"\x09"]
    assert abs(got[1][0] - 1567108497.999999) < 1e-6


def test_btle_scan_stop_rules_without_gpu(monkeypatch):
    """check_stop semantics of snout/util/btle.py:111-122 on a stubbed line source."""
    from snout_amd.scan import BtleScan
    line = (b"1567108496.651985 Pkt8 Ch37 AA:8e89bed6 ADV_PDU_t0:ADV_IND T1 R0 PloadL20 "
            b"AdvA:6385725ebfcd Data:0201060aff4c001005011c569415 CRC0\n")
    scan = BtleScan(channels=[37, 38], source=None, timeout=None, packet_threshold=5)

    def fake_lines(ch):
        for i in range(4):
            scan._elapsed = i * 0.1
            yield line if i % 2 == 0 else line.replace(b"CRC0", b"CRC1")
    monkeypatch.setattr(scan, "lines", fake_lines)
    seen = []
    scan.events.on("btle.packet-received", lambda message: seen.append(message))
    msgs = scan.run()
    assert len(msgs) == 4 == len(seen)          # CRC1 lines dropped (message.py:226), 2 per channel
    scan2 = BtleScan(channels=[37], source=None, timeout=0.15)
    monkeypatch.setattr(scan2, "lines", lambda ch: ((setattr(scan2, "_elapsed", i * 0.1), line)[1]
                                                     for i in range(10)))
    assert len(scan2.run()) == 3                # stops once capture time >= timeout
