"""Host logic that needs no GPU: channel parsing, pcap writer, scan stop rules with a fake source."""
import numpy as np
import pytest

from snout_amd import cli, formats


def test_parse_channels_like_reference():
    """Grammar of ChannelsOption + get_channels (snout/util/iot_click.py:46-92, snout/cli.py:139-183)."""
    assert cli.parse_channels("37", "btle") == [37]
    assert cli.parse_channels("37,38,39", "btle") == [37, 38, 39]          # literal tuple -> as given
    assert cli.parse_channels("[11,12,13]", "zigbee") == [11, 12, 13]
    assert cli.parse_channels("[15]", "zigbee") == [15, 15]                # upstream doubles a one-element list
    assert cli.parse_channels("11:14", "zigbee") == [11, 12, 13, 14]
    assert cli.parse_channels("11-14", "zigbee") == [11, 12, 13, 14]
    assert cli.parse_channels("all", "zigbee") == list(range(11, 27))
    assert cli.parse_channels("ALL", "btle") == list(range(0, 40))
    assert cli.parse_channels("", "zigbee") == [11] and cli.parse_channels(None, "btle") == [37]
    for bad, proto in (("40", "btle"), ("10", "zigbee"), ("11:27", "zigbee"), ("x", "btle"), ("{1:2}", "btle")):
        with pytest.raises(Exception):
            cli.parse_channels(bad, proto)


def test_stop_conditions_defaults(monkeypatch):
    """snout/cli.py:246-250: with neither -n nor -t the reference asks for both; without a terminal the
    prompts' defaults apply (no packet threshold, the protocol's timeout of snout/util/__init__.py:4-9)."""
    import io
    monkeypatch.setattr("sys.stdin", io.StringIO(""))
    assert cli.stop_conditions("btle", None, None) == (None, 10.0)
    assert cli.stop_conditions("zigbee", 5, None) == (5, None)
    assert cli.stop_conditions("zigbee", None, 2.5) == (None, 2.5)
    assert cli.stop_conditions("btle", 0, None) == (None, None)            # --num 0: unlimited, as upstream
    with pytest.raises(Exception):
        cli.stop_conditions("btle", -1, None)


def test_console_entries_and_option_names():
    """`snout {btle,zigbee} scan` with the reference's option names (-c/--channels, -a/--active, -n/--num,
    -t/--timeout: snout/cli.py:220-236), the in-tree entry scripts and setup.py's console_scripts."""
    import os
    import subprocess
    import sys
    from click.testing import CliRunner
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for proto in ("btle", "zigbee"):
        out = CliRunner().invoke(cli.main, [proto, "scan", "--help"]).output
        for opt in ("-c, --channels", "-a, --active", "-n, --num", "-t, --timeout", "--wideband", "--sharded"):
            assert opt in out, (proto, opt)
    r = subprocess.run([sys.executable, os.path.join(root, "bin", "snout"), "--help"], capture_output=True, timeout=120)
    assert r.returncode == 0 and b"btle" in r.stdout and b"zigbee" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bin", "btle_rx"), "--help"], capture_output=True, timeout=120)
    assert r.returncode == 0 and b"-c" in r.stdout and b"-k" in r.stdout
    setup = open(os.path.join(root, "setup.py")).read()
    assert "snout = snout_amd.cli:main" in setup and "btle_rx = snout_amd.cli:btle_rx_main" in setup


def test_pcap_roundtrip(tmp_path):
    p = str(tmp_path / "z.pcap")
    frames = [(1567108496.25, bytes.fromhex("03083affffffff07aabb")), (1567108497.999999, b"\x01\x02\x03")]
    assert formats.write_pcap(p, frames) == 2
    assert formats.write_pcap(p, [(1567108498.5, b"\x09")], append=True) == 1
    lt, got = formats.read_pcap(p)
    assert lt == 195 and [g[1] for g in got] == [f[1] for f in frames] + [b"\x09"]
    assert abs(got[1][0] - 1567108497.999999) < 1e-6


def test_pcap_bytes_are_the_libpcap_file_format(tmp_path):
    """Byte-level known answer, independent of `read_pcap`: what `scapy.wrpcap(filename, pkt.payload, append=True)`
    (snout/util/zigbee.py:202) puts on disk for 802.15.4 frames is the classic libpcap file -- global header: magic
    a1b2c3d4 in the writer's byte order, version 2.4, thiszone 0, sigfigs 0, snaplen 65535, link type 195
    (LINKTYPE_IEEE802_15_4_WITHFCS); per frame: seconds, microseconds, captured length, wire length, then the bytes --
    and `file(1)` recognises it."""
    import subprocess
    p = str(tmp_path / "k.pcap")
    mpdu = bytes.fromhex("03083affffffff07aabb")
    assert formats.write_pcap(p, [(1567108496.25, mpdu)]) == 1
    raw = open(p, "rb").read()
    assert raw[:24] == bytes.fromhex("d4c3b2a1" "0200" "0400" "00000000" "00000000" "ffff0000" "c3000000")
    assert raw[24:40] == (1567108496).to_bytes(4, "little") + (250000).to_bytes(4, "little") + (10).to_bytes(4, "little") * 2
    assert raw[40:] == mpdu and len(raw) == 50
    # appending keeps ONE global header
    assert formats.write_pcap(p, [(1567108497.0, b"\x01\x02")], append=True) == 1
    raw2 = open(p, "rb").read()
    assert raw2[:50] == raw and raw2.count(bytes.fromhex("d4c3b2a1")) == 1 and len(raw2) == 50 + 16 + 2
    out = subprocess.run(["file", "-b", p], capture_output=True, text=True).stdout.lower()
    assert "pcap capture file" in out and "version 2.4" in out and ("802.15.4" in out or "195" in out), out


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


def test_wideband_scans_refuse_live_streams(tmp_path):
    """`--iq -` / a FIFO is a live stream, implemented for the single-channel scans; with --wideband / --sharded the
    value used to reach WidebandSource, which opened a file literally named '-'.  Now a usage error, before anything touches
    the GPU."""
    import os
    from click.testing import CliRunner
    from snout_amd import cli
    fifo = str(tmp_path / "iq.fifo")
    os.mkfifo(fifo)
    for proto in ("btle", "zigbee"):
        for how in ("--wideband", "--sharded"):
            for iq in ("-", fifo):
                r = CliRunner().invoke(cli.main, [proto, "scan", how, "--iq", iq, "-c", "all", "-t", "1"])
                assert r.exit_code == 2 and "capture FILE" in r.output, (proto, how, iq, r.output)
