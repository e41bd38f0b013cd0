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


# ---- IEEE 802.15.4 MAC header ------------------------------------------------
FRAME_TYPES = {0: "Beacon", 1: "Data", 2: "Ack", 3: "Command"}


def parse_mhr(mpdu: bytes) -> dict:
    """Minimal 802.15.4-2006 MAC header parse of an MPDU (FCS included): frame control fields,
    sequence number, destination / source PAN and address — what the reference reads off scapy's
    ``Dot15d4FCS`` (snout/util/zigbee.py:14,197).  Addresses are returned as integers (short: 16 bit,
    extended: 64 bit), absent fields as None; ``header_len`` is the offset of the MAC payload.
    Raises ValueError on a frame too short for the fields its frame control announces."""
    if len(mpdu) < 3:
        raise ValueError("MPDU shorter than frame control + sequence number")
    fc = mpdu[0] | (mpdu[1] << 8)
    out = {
        "frame_type": fc & 7, "frame_type_name": FRAME_TYPES.get(fc & 7, "Reserved"),
        "security": bool(fc >> 3 & 1), "pending": bool(fc >> 4 & 1), "ack_request": bool(fc >> 5 & 1),
        "panid_compress": bool(fc >> 6 & 1), "dest_mode": fc >> 10 & 3, "version": fc >> 12 & 3,
        "src_mode": fc >> 14 & 3, "seq": mpdu[2],
        "dest_pan": None, "dest_addr": None, "src_pan": None, "src_addr": None,
    }
    pos = 3

    def take(n):
        nonlocal pos
        if pos + n > len(mpdu):
            raise ValueError("MPDU ends inside the addressing fields")
        v = int.from_bytes(mpdu[pos:pos + n], "little")
        pos += n
        return v

    if out["dest_mode"] in (2, 3):
        out["dest_pan"] = take(2)
        out["dest_addr"] = take(2 if out["dest_mode"] == 2 else 8)
    if out["src_mode"] in (2, 3):
        out["src_pan"] = out["dest_pan"] if (out["panid_compress"] and out["dest_pan"] is not None) else take(2)
        out["src_addr"] = take(2 if out["src_mode"] == 2 else 8)
    out["header_len"] = pos
    return out
