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


_AGO_UNITS = (60.0, 60.0, 24.0, 7.0, 365.0 / 7.0 / 12.0, 12.0)
_AGO_TEXT = ("just now", "%s seconds ago", "1 minute ago", "%s minutes ago", "1 hour ago", "%s hours ago", "1 day ago",
             "%s days ago", "1 week ago", "%s weeks ago", "1 month ago", "%s months ago", "1 year ago", "%s years ago")


def _ago(seconds: float) -> str:
    """English text of the `timeago` package the reference formats its times with (device.py:139-143, :279): the
    difference is divided through 60 s, 60 min, 24 h, 7 d, weeks per month, 12 months while it reaches the next unit;
    "just now" up to 9 s, the singular form at exactly one unit.  Pinned by tests/golden/devices.json."""
    diff = max(0.0, float(seconds))
    i = 0
    while i < len(_AGO_UNITS) and diff >= _AGO_UNITS[i]:
        diff /= _AGO_UNITS[i]
        i += 1
    n = int(diff)
    i *= 2
    if n > (9 if i == 0 else 1):
        i += 1
    t = _AGO_TEXT[i]
    return t % n if "%s" in t else t


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
        """The "Up" column exactly as the reference prints it (device.py:155-167) -- including its arithmetic above one
        hour, which subtracts the HOUR COUNT (not the hours' seconds) before taking minutes and seconds: 3725 s prints as
        ``01:62:04``, one hour sharp as ``60:00``.  A drop-in prints what the reference prints; :attr:`uptime_hms` is the
        same duration in conventional clock form."""
        up = self.uptime
        if up < 0:
            return "-"
        nice = "%02d:%02d" % (up / 60, up % 60)
        if up > 60 * 60:
            hours = int(up / (60 * 60))
            nice = "%02d:%02d:%02d" % (up / (60 * 60), (up - hours) / 60, (up - hours) % 60)
        return nice

    @property
    def uptime_hms(self) -> str:
        """hh:mm:ss (mm:ss below one hour) of :attr:`uptime`; not a column of the reference's table."""
        up = self.uptime
        if up < 0:
            return "-"
        if up >= 3600:
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
