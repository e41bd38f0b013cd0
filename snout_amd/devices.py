"""Device table of a BTLE scan and the `.b` line dump.

Counterparts in the reference: ``Device`` (snout/core/device.py:131-295: last_seen, occurrences,
uptime, vendor / model / os / activity read from the dissected AdvData of the messages a device
sent), the summary table of ``BtleScanUIHandlerSummary`` (snout/util/btle.py:202-240: columns MAC,
Last Seen, #, Up, Vendor, Model, OS, Info, most recent first, at most 51 rows) and the `.b` dump
(snout/util/btle.py:46,105-106: every accepted btle_rx line appended verbatim).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional

from .message import BtleMessage

FITBIT_UUID128 = "ba5689a6fabfa2bd01467d6e00fbabad"
UUID128_KEY = "Incomplete List of 128-bit Service Class UUIDs"
COLUMNS = ["MAC", "Last Seen", "#", "Up", "Vendor", "Model", "OS", "Info"]


def _ago(seconds: float) -> str:
    s = int(max(0.0, seconds))
    if s < 60:
        return "just now" if s < 10 else f"{s} seconds ago"
    if s < 3600:
        return f"{s // 60} minute{'s' if s >= 120 else ''} ago"
    return f"{s // 3600} hour{'s' if s >= 7200 else ''} ago"


@dataclass
class DeviceEntry:
    id: str
    messages_sent: List[BtleMessage] = field(default_factory=list)
    _dissected: List[dict] = field(default_factory=list)

    def add(self, m: BtleMessage) -> None:
        self.messages_sent.append(m)
        try:
            self._dissected.append(m.payload)
        except (IndexError, ValueError):        # malformed AdvData: counted, not dissected
            self._dissected.append({})

    @property
    def last_seen(self) -> float:
        return self.messages_sent[-1].timestamp if self.messages_sent else 0

    @property
    def occurrences(self) -> int:
        return len(self.messages_sent)

    @property
    def uptime(self) -> int:
        if not self.messages_sent:
            return -1
        return round(self.messages_sent[-1].timestamp - self.messages_sent[0].timestamp)

    @property
    def uptime_nice(self) -> str:
        up = self.uptime
        if up < 0:
            return "-"
        if up > 3600:
            return "%02d:%02d:%02d" % (up // 3600, (up % 3600) // 60, up % 60)
        return "%02d:%02d" % (up // 60, up % 60)

    def _fitbit(self, d: dict) -> bool:
        return FITBIT_UUID128 in (d.get(UUID128_KEY) or "")

    @property
    def vendor(self) -> str:
        for d in self._dissected:
            if d.get("company_name"):
                return d["company_name"]
            if self._fitbit(d):
                return "FitBit"
        return "-"

    @property
    def model(self) -> str:
        for d in self._dissected:
            if self._fitbit(d):
                return "Charge / Charge HR"
            recs = d.get("manufacturer-specific")
            if d.get("company_id") == 0x004C and isinstance(recs, list):
                if any(isinstance(r, dict) and r.get("type") == "AirPods" for r in recs):
                    return "AirPods"
        return "-"

    @property
    def os(self) -> str:
        for d in self._dissected:
            recs = d.get("manufacturer-specific")
            if d.get("company_id") == 0x004C and isinstance(recs, list):
                for r in recs:
                    if isinstance(r, dict) and r.get("type") == "Nearby":
                        hint = r.get("iOS Version Hint")
                        return "iOS " + hint if hint else "-"
            if d.get("company_id") == 0x0006:
                return "Windows 10 >= v10.0.10240.0"
        return "-"

    def activity(self, now: float) -> str:
        """The last (at most three) changes of Apple's Nearby action, oldest first."""
        seq = []
        for m, d in zip(self.messages_sent, self._dissected):
            recs = d.get("manufacturer-specific")
            if isinstance(recs, list):
                for r in recs:
                    if isinstance(r, dict) and r.get("type") == "Nearby" and r.get("Action Code Text"):
                        seq.append((m.timestamp, r["Action Code Text"]))
        changes, prev = [], None
        for t, a in seq:
            if a != prev:
                changes.append((t, a))
                prev = a
        changes = changes[-3:]
        return ", ".join(f"{_ago(now - t)}: {a}" for t, a in changes) if changes else "-"


class DeviceTable:
    """Devices keyed by advertiser address, fed with the messages of a scan."""

    def __init__(self):
        self.devices: Dict[str, DeviceEntry] = {}

    def add(self, m: BtleMessage) -> DeviceEntry:
        dev = self.devices.get(m.sender)
        if dev is None:
            dev = self.devices[m.sender] = DeviceEntry(m.sender)
        dev.add(m)
        return dev

    def extend(self, messages: Iterable[BtleMessage]) -> "DeviceTable":
        for m in messages:
            self.add(m)
        return self

    def rows(self, now: Optional[float] = None, limit: int = 51) -> List[list]:
        """Rows of the summary table, most recently seen device first."""
        devs = sorted((d for d in self.devices.values() if d.messages_sent),
                      key=lambda d: d.last_seen, reverse=True)
        now = devs[0].last_seen if (now is None and devs) else (now or 0.0)
        return [[d.id, _ago(now - d.last_seen), d.occurrences, d.uptime_nice, d.vendor, d.model, d.os,
                 d.activity(now)] for d in devs[:limit]]

    def render(self, now: Optional[float] = None) -> str:
        from tabulate import tabulate
        return tabulate(self.rows(now), headers=COLUMNS, tablefmt="pretty")


def read_b_dump(path: str) -> List[BtleMessage]:
    """Messages of a `.b` dump: one btle_rx line per accepted packet, as BtleScan writes them."""
    out = []
    with open(path, "rb") as f:
        for line in f:
            m = BtleMessage.fromraw(line)
            if m:
                out.append(m)
    return out
