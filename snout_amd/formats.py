"""On-disk formats next to the path: the `.b` line dump is written by BtleScan
itself (snout/util/btle.py:46,105-106); this module writes the pcap the reference gets from
``scapy.wrpcap(filename, pkt.payload, append=True)`` (snout/util/zigbee.py:202) — link type 195
(IEEE 802.15.4 with FCS) — without importing scapy."""
from __future__ import annotations

import struct
from typing import Iterable, Tuple

LINKTYPE_IEEE802_15_4_WITHFCS = 195


def write_pcap(path: str, frames: Iterable[Tuple[float, bytes]], linktype: int = LINKTYPE_IEEE802_15_4_WITHFCS,
               append: bool = False) -> int:
    """Classic pcap (magic a1b2c3d4, v2.4, microsecond timestamps). Returns frames written."""
    import os
    n = 0
    new = not (append and os.path.exists(path) and os.path.getsize(path) >= 24)
    with open(path, "wb" if new else "ab") as f:
        if new:
            f.write(struct.pack("<IHHiIII", 0xA1B2C3D4, 2, 4, 0, 0, 65535, linktype))
        for ts, data in frames:
            sec = int(ts)
            usec = int(round((ts - sec) * 1e6))
            if usec >= 1000000:
                sec, usec = sec + 1, usec - 1000000
            f.write(struct.pack("<IIII", sec, usec, len(data), len(data)))
            f.write(data)
            n += 1
    return n


def read_pcap(path: str):
    """-> (linktype, [(timestamp, bytes)])"""
    with open(path, "rb") as f:
        magic, _, _, _, _, _, lt = struct.unpack("<IHHiIII", f.read(24))
        assert magic == 0xA1B2C3D4
        out = []
        while True:
            h = f.read(16)
            if len(h) < 16:
                break
            sec, usec, incl, _ = struct.unpack("<IIII", h)
            out.append((sec + usec / 1e6, f.read(incl)))
    return lt, out
