"""GPU: the scan surface around the receive path — the UDP hand-over to scapy-radio
(Zigbee_rx/top_block.py:71 `socket_pdu("UDP_CLIENT", '127.0.0.1', '52002', ...)`, consumer
snout/core/radio.py:232-237), the wideband / sharded modes of `snout {btle,zigbee} scan`."""
import json
import os
import socket
import struct
import subprocess
import sys

import numpy as np
import pytest

from snout_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_udp_52002_receives_what_the_flowgraph_would_send():
    """Bind the port scapy-radio's GnuradioSocket listens on, run ZigbeeScan(udp=True) and check every
    datagram: RFtap header (magic, len32 = 4, flags 0x81, DLT 195, qual = lqi / 255) + MPDU incl. FCS."""
    from snout_amd.scan import ArraySource, ZigbeeScan
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    try:
        rx.bind(("127.0.0.1", 52002))
    except OSError:
        pytest.skip("UDP 52002 is taken on this box")
    rx.settimeout(5.0)
    try:
        x, truth = synth.zigbee_capture(1 << 19, channel=15, seed=31, mean_gap=15000.0)
        scan = ZigbeeScan(channels=[15], source=ArraySource({15: x}), timeout=None, udp=True)
        msgs = scan.run()
        got = []
        for _ in msgs:
            got.append(rx.recvfrom(4096)[0])
    finally:
        rx.close()
    assert len(got) == len(msgs) >= len(truth) > 5
    sent = {t.payload for t in truth}
    for d, m in zip(got, msgs):
        assert d == m.datagram and d[:4] == b"RFta"
        len32, flags, dlt = struct.unpack("<HHI", d[4:12])
        assert (len32, flags, dlt) == (4, 0x0081, 195)
        assert abs(struct.unpack("<f", d[12:16])[0] - m.lqi / 255.0) < 1e-6
        assert d[16:] == m.mpdu
    assert sum(1 for m in msgs if m.mpdu in sent) == len(truth)


def test_wideband_source_equals_the_oracle(oracle):
    """BtleScan over a whole-band capture (WidebandSource -> ShardedScan -> channelizer handle, several
    overlapping segments) reports what the oracle's wideband receiver finds in one piece."""
    from snout_amd.scan import BtleScan, WidebandSource
    x, truth = synth.wideband_capture(0, 40 * 60000, seed=9, mean_gap=9000.0)
    src = WidebandSource(x, 0, segment=40 * 20000)
    chans = sorted({t.channel for t in truth})
    scan = BtleScan(channels=chans, source=src, timeout=None, t0_epoch=0.0)
    lines = []
    scan.events.on("btle.packet-received", lambda message: lines.append(message))
    msgs = scan.run()
    want = oracle.wideband_segment(x, proto=0)
    want_ok = want[want["crc_ok"] == 1]
    assert len(msgs) == len(want_ok) >= 0.9 * len(truth)
    assert sorted(int(m.channel) for m in msgs) == sorted(int(c) for c in want_ok["channel"])
    sent = {t.payload[2:8][::-1].hex() for t in truth}
    assert {m.sender for m in msgs} <= sent


def test_cli_wideband_scan_all_channels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "snout"), "btle", "scan", "--wideband", "--synthetic",
                        "--seconds", "0.02", "-c", "all", "-t", "1"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    from snout_amd.message import BtleMessage
    msgs = [m for m in (BtleMessage.fromraw(ln) for ln in r.stdout.splitlines(keepends=True)) if m]
    assert len(msgs) >= 20 and len({m.channel for m in msgs}) >= 10
    assert all(m.access_address == "8e89bed6" for m in msgs)


def test_cli_sharded_two_ranks_over_gloo(tmp_path):
    """`--sharded` under torch.distributed.run: segments dealt round-robin, records gathered on rank 0,
    the same frames as one rank finds."""
    x, truth = synth.wideband_capture(1, 16 * 150000, seed=4, bins=range(0, 16, 2), max_len=60)
    path = str(tmp_path / "zb_wide.cf32")
    x.tofile(path)
    cmd = [os.path.join(ROOT, "bin", "snout"), "zigbee", "scan", "--sharded", "--iq", path, "--segment", str(16 * 50000),
           "-c", "all", "-t", "5"]
    one = subprocess.run([sys.executable] + cmd, capture_output=True, timeout=600)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SNOUT_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port)] + cmd,
                         capture_output=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr.decode()[-3000:]
    f1 = sorted(ln.split()[-1] for ln in one.stdout.decode().splitlines() if " Ch" in ln)
    f2 = sorted(ln.split()[-1] for ln in two.stdout.decode().splitlines() if " Ch" in ln)
    assert len(f1) >= 0.8 * len(truth) and f1 == f2


def test_gpu_scan_feeds_the_reference_device_table():
    """IQ -> `BtleScan` on the GPU -> btle_rx lines -> AdvData dissection -> device table, end to end against the
    REFERENCE's own classes: tests/golden/devices.json holds, for the "capture" scenario, the lines the CPU oracle prints
    for a capture of known advertising PDUs and the table rows the imported reference (snout/core/message.py:205-237,
    snout/core/device.py:131-295, snout/util/btle.py:202-240) makes of them (tests/golden/make_golden_devices.py).  The
    GPU scan of the same capture must print the same lines and fill the same table."""
    from snout_amd import devices
    from snout_amd.scan import ArraySource, BtleScan
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "devices.json")))["scenarios"]["capture"]
    cap = gold["capture"]
    pdus = [synth.btle_adv_pdu(0, bytes.fromhex(mac), bytes.fromhex(adv), txadd=1) for mac, adv in cap["packets"]]
    x, _ = synth.btle_capture_of(pdus, spacing=cap["spacing"], seed=cap["seed"], sigma=cap["sigma"])
    scan = BtleScan(channels=[37], source=ArraySource({37: x}), timeout=None, t0_epoch=cap["t0"])
    lines = [ln.decode() for ln in scan.lines(37)]
    assert lines == gold["lines"]
    scan2 = BtleScan(channels=[37], source=ArraySource({37: x}), timeout=None, t0_epoch=cap["t0"])
    msgs = scan2.run()
    assert len(msgs) == len(pdus) == sum(gold["accepted"])
    table = devices.DeviceTable().extend(msgs)
    assert table.rows(now=gold["now"]) == gold["rows"]
    # the dissected payload of a decoded packet is what the reference's dissector returns for its AdvData (advdata.json pins
    # the dissector itself): Apple Nearby, action 3, iOS 12 hint, Wi-Fi on
    first = msgs[0].payload
    assert first["company_id"] == 0x004C and first["manufacturer-specific"][0]["Action Code Text"] == "Locked Screen"
    assert first["manufacturer-specific"][0]["iOS Version Hint"] == "12" and first["manufacturer-specific"][0]["Wi-Fi"] == "On"


def _feed(path, data: bytes, chunk: int, delay: float):
    """Writer side of a live stream: the capture in paced chunks, as a radio's transfers arrive."""
    import time
    with open(path, "wb", buffering=0) as f:
        for lo in range(0, len(data), chunk):
            f.write(data[lo:lo + chunk])
            time.sleep(delay)


@pytest.mark.parametrize("fmt", [0, 1])
def test_live_stream_gives_the_records_of_the_file(tmp_path, fmt):
    """Row a11's live half: `StreamSource` reads a FIFO fed in paced chunks (odd sizes: segment boundaries fall anywhere),
    cuts overlapping segments into submit / collect and must print exactly the lines of the same samples read from a
    file -- cf32 and the HackRF's int8 pairs; the wall clock, not capture time, is what `timeout` measures."""
    import threading
    import time
    from snout_amd.scan import BtleScan, FileSource, StreamSource
    x, truth = synth.btle_capture(3 * (1 << 20) + 12345, channel=37, seed=77, mean_gap=6000.0)
    if fmt == 1:
        data = synth.quantize(x, 1).tobytes()
    else:
        data = x.tobytes()
    f = tmp_path / "cap.bin"
    f.write_bytes(data)
    want = list(BtleScan(channels=[37], source=FileSource(str(f), fmt), timeout=None, t0_epoch=100.0).lines(37))
    assert len(want) >= len(truth) > 300
    fifo = str(tmp_path / "iq.fifo")
    os.mkfifo(fifo)
    th = threading.Thread(target=_feed, args=(fifo, data, 999983, 0.002))
    th.start()
    src = StreamSource(fifo, fmt, segment=1 << 19)
    got = list(BtleScan(channels=[37], source=src, timeout=None, t0_epoch=100.0).lines(37))
    th.join()
    assert src.samples_read == len(x) and got == want
    # wall-clock timeout: a slow stream is left after `timeout` seconds although the capture time is far shorter
    os.unlink(fifo)
    os.mkfifo(fifo)
    slow = data[:(len(data) // 6) // 8 * 8]
    bps = 8 if fmt == 0 else 2
    th = threading.Thread(target=_feed, args=(fifo, slow, (1 << 15) * bps, 0.2))       # 2^15 samples every 0.2 s
    th.start()
    t0 = time.time()
    scan = BtleScan(channels=[37], source=StreamSource(fifo, fmt, segment=1 << 17), timeout=1.0, t0_epoch=100.0)
    try:
        msgs = scan.run()
        took = time.time() - t0
    finally:
        # unblock the writer: read the rest of the FIFO away
        with open(fifo, "rb") as rest:
            while rest.read(1 << 20):
                pass
        th.join()
    # 2^17-sample segments arrive every 0.8 s and the first is collected once the second is on its way: the scan ends with
    # the first records after 1 s, long before the stream does (16 chunks = 3.2 s) and with a fraction of its packets
    assert 0 < len(msgs) < len(want) // 8 and 1.0 <= took < 3.0


def test_live_stream_zigbee_and_the_cli_on_stdin(tmp_path):
    """802.15.4 over a live stream (segments restart the DC filter: every later segment starts a pre-roll early and leaves
    what it finds there to the segment before), and the drop-in child on a pipe: `... | btle_rx -c 37 --iq -`."""
    import threading
    from snout_amd.scan import StreamSource, ZigbeeScan
    x, truth = synth.zigbee_capture(1 << 21, channel=15, seed=78, mean_gap=20000.0)
    fifo = str(tmp_path / "zb.fifo")
    os.mkfifo(fifo)
    th = threading.Thread(target=_feed, args=(fifo, x.tobytes(), 1 << 20, 0.001))
    th.start()
    msgs = ZigbeeScan(channels=[15], source=StreamSource(fifo, 0, segment=1 << 19), timeout=None).run()
    th.join()
    sent = [t.payload for t in truth]
    found = [m.mpdu for m in msgs]
    assert len(truth) > 20 and sum(p in found for p in sent) >= len(sent) - 1
    assert len(found) == len(set((m.timestamp, m.mpdu) for m in msgs))              # nothing reported twice
    # the child process on stdin
    xb, tb = synth.btle_capture(1 << 20, channel=37, seed=79, mean_gap=8000.0)
    f = tmp_path / "b.cf32"
    xb.tofile(f)
    exe = [sys.executable, os.path.join(ROOT, "bin", "btle_rx"), "-c", "37", "-g", "6", "-a", "8e89bed6", "-k", "555555"]
    ref = subprocess.run(exe + ["--iq", str(f)], capture_output=True, timeout=300)
    piped = subprocess.run(exe + ["--iq", "-"], input=xb.tobytes(), capture_output=True, timeout=300)
    assert ref.returncode == 0 and piped.returncode == 0, piped.stderr.decode()[-2000:]

    def strip_time(out):        # the epoch of a line is the process's start time + sample_index / fs
        return [ln.split(b" ", 1)[1] for ln in out.splitlines()]
    assert len(strip_time(ref.stdout)) >= len(tb) > 50 and strip_time(piped.stdout) == strip_time(ref.stdout)
