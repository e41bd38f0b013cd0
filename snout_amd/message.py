"""Packet-model counterpart of snout/core/message.py for the receive path.

``BtleMessage.fromraw`` keeps the reference's contract (snout/core/message.py:205-237): the raw
message is one ``btle_rx`` stdout line; it is accepted only if it has exactly 11 space-separated
tokens and the last one is ``'CRC0\\n'`` (:226); the fields are cut by fixed slices (:228-236).
Like the reference, a rejected line returns ``False``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional


@dataclass
class BtleMessage:
    sender: str
    channel: str
    pdu_type: str
    payload_hex: str
    timestamp: float
    number: int
    raw: bytes = b""
    access_address: str = "8e89bed6"
    protocol: str = "btle"
    meta: dict = field(default_factory=dict)

    @property
    def payload(self) -> dict:
        """Dissected AdvData (BtlePDUPayload.fromstring(...).dict() in the reference, message.py:232)."""
        from .advertising import dissect_hex
        return dissect_hex(self.payload_hex)

    @classmethod
    def fromraw(cls, raw_message: Optional[bytes]):
        if not raw_message:
            return False
        tok = raw_message.decode().split(" ")
        if tok[-1] != "CRC0\n" or len(tok) != 11:
            return False
        return cls(sender=tok[8][5:], channel=tok[2][2:], pdu_type=tok[4][11:],
                   payload_hex=tok[9][5:], timestamp=float(tok[0]), number=int(tok[1][3:]),
                   raw=raw_message)


@dataclass
class ZigbeeMessage:
    """One 802.15.4 frame as delivered on the scapy-radio surface: RFtap header + MPDU
    (top_block.py:53,71; consumer snout/util/zigbee.py:194-202)."""
    channel: int
    mpdu: bytes
    lqi: int
    qual: float
    timestamp: float
    datagram: bytes = b""
    protocol: str = "zigbee"
