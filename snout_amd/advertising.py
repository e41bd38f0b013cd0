"""BTLE AdvData dissection — the consumer right after the receive path.

Counterpart of ``BtlePDUPayload`` (snout/core/protocols/btle/advertising.py:113-307): the hex string
of token 9 of a btle_rx line (``Data:...``, message.py:232) becomes a dict with the same keys and
value conventions, pinned by tests/golden/advdata.json (produced by importing the reference).
Quirks kept on purpose so outputs stay comparable: 16-bit little-endian words are what the
reference calls ``word16be``; a later AD structure of the same kind overwrites an earlier one;
an Apple Handoff/Nearby record shorter than its fixed fields raises IndexError.
Only a handful of company names are carried (the reference ships the full Bluetooth SIG list,
snout/core/protocols/btle/assigned_numbers/company_ids.py); unknown ids give '??' like its
``dict.get`` fallback.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Tuple

FLAG_BITS = [
    "LE Limited Discoverable Mode",
    "LE General Discoverable Mode",
    "BR/EDR Not Supported (i.e. bit 37 of LMP Extended Feature bits Page 0)",
    "Simultaneous LE and BR/EDR to Same Device Capable (Controller) (i.e. bit 49 of LMP Extended Feature bits Page 0)",
    "Simultaneous LE and BR/EDR to Same Device Capable (Host) (i.e. bit 66 of LMP Extended Feature bits Page 1)",
]
OOB = ("OOB data present", "OOB data not present")
OOB_LE = "LE supported (Host) (i.e. bit 65 of LMP Extended Feature bits Page 1"
OOB_LE_BR = "Simultaneous LE and BR/EDR to Same Device Capable (Host) (i.e. bit 66 of LMP Extended Fea- ture bits Page 1"
OOB_ADDR = ("Address Type: Random Address", "Address Type: Public Address")

AD_TYPE_NAMES = {0x06: "Incomplete List of 128-bit Service Class UUIDs"}
COMPANIES = {0x0006: "Microsoft", 0x004C: "Apple, Inc.", 0x0075: "Samsung Electronics Co. Ltd.",
             0x00E0: "Google", 0x0059: "Nordic Semiconductor ASA"}
APPLE_TYPES = {0x02: "iBeacon", 0x05: "AirDrop", 0x07: "AirPods", 0x09: "AirPlay Destination",
               0x0A: "AirPlay Source", 0x0C: "Handoff", 0x0D: "Wi-Fi Settings", 0x0E: "Instant Hotspot",
               0x0F: "Wi-Fi Join Network", 0x10: "Nearby"}
APPLE_ACTIONS = {1: "iOS recently updated", 3: "Locked Screen", 7: "Transition Phase",
                 10: "Locked Screen, Inform Apple Watch", 11: "Active User", 13: "Unknown",
                 14: "Phone Call or Facetime"}


def _le16(b: bytes) -> int:
    return (b[1] << 8) | b[0]


def ad_structures(adv: bytes) -> Iterator[bytes]:
    """(length, data) fields of AdvData; a truncated last field yields what is there."""
    pos = 0
    while pos < len(adv):
        n = adv[pos]
        yield adv[pos + 1:pos + 1 + n]
        pos += 1 + n


def _tlv(buf: bytes) -> Iterator[Tuple[int, bytes]]:
    """Apple's (type, length, data) records."""
    pos = 0
    while pos < len(buf):
        t, n = buf[pos], buf[pos + 1]
        yield t, buf[pos + 2:pos + 2 + n]
        pos += 2 + n


def _apple(man: bytes) -> List[dict]:
    out = []
    for t, d in _tlv(man):
        if t not in APPLE_TYPES:
            out.append({"type": hex(t), "data": d.hex()})
            continue
        rec: Dict[str, object] = {"type": APPLE_TYPES[t]}
        if t == 0x0C:
            rec["Clipboard Status"] = d[0]
            rec["Sequence Number"] = _le16(d[1:3])
        elif t == 0x0D:
            rec["iCloud ID"] = d[2:].hex()
        elif t == 0x0E:
            for key, i in (("Battery Life", 4), ("Cell Service", 6), ("Cell Bars", 7)):
                if i < len(d):
                    rec[key] = d[i]
        elif t == 0x0F:
            rec["data"] = d.hex()
        elif t == 0x10:
            rec["Location Sharing"] = d[0] >> 4
            rec["Action Code"] = d[0] & 0x0F
            rec["Action Code Text"] = APPLE_ACTIONS.get(d[0] & 0x0F, "??")
            nb = d[1:]
            if len(nb) == 1 and nb[0] == 0x00:
                rec["iOS Version Hint"] = "10"
            if len(nb) == 4:
                rec["Data"] = nb[1:]
                if nb[0] == 0x10:
                    rec["iOS Version Hint"] = "11"
                if nb[0] in (0x18, 0x1C):
                    rec["iOS Version Hint"] = "12"
                    rec["Wi-Fi"] = "On" if nb[0] == 0x1C else "Off"
        out.append(rec)
    return out


def dissect(adv: bytes) -> dict:
    """AdvData bytes -> dict (keys: hex, flags, sec-mg-oob-flags, service-data, company_id,
    company_raw, company_name, manufacturer-specific, unknown, <AD type name>)."""
    d: Dict[str, object] = {"hex": adv.hex()}
    for s in ad_structures(adv):
        if not s:
            continue
        t, body = s[0], s[1:]
        if t == 0x01:
            d["flags"] = [name for bit, name in enumerate(FLAG_BITS) if body[0] >> bit & 1]
        elif t == 0x06:
            d[AD_TYPE_NAMES[0x06]] = body.hex()
        elif t == 0x11:
            v = body[0]
            fl = [OOB[0] if v & 1 else OOB[1]]
            if v & 2:
                fl.append(OOB_LE)
            if v & 4:
                fl.append(OOB_LE_BR)
            fl.append(OOB_ADDR[0] if v & 8 else OOB_ADDR[1])
            d["sec-mg-oob-flags"] = fl
        elif t == 0x16:
            d["service-data"] = {"uuid": _le16(body[0:2]), "data": body[2:]}
        elif t == 0xFF:
            cid, man = _le16(body[0:2]), body[2:]
            d["company_id"] = cid
            d["company_raw"] = man.hex()
            d["company_name"] = COMPANIES.get(cid, "??")
            if cid == 0x004C:
                d["manufacturer-specific"] = _apple(man)
            elif cid == 0x0006:
                d["manufacturer-specific"] = man.hex()
        else:
            d["unknown"] = {"type": t, "hex": body.hex()}
    return d


def dissect_hex(hexstr: str) -> dict:
    return dissect(bytes.fromhex(hexstr))
