"""`snout {btle,zigbee} scan` — the CLI surface of the reference (snout/cli.py:41-56,220-261) for the
receive path: same option names (-c channels, -n packets, -t timeout, -f filename), samples taken
from a recorded cf32 file (--iq) or a synthetic capture (--synthetic) instead of a live SDR.

`snout-rx btle-rx -c 37 -a 8e89bed6 -k 555555 --iq FILE` prints btle_rx-format lines on stdout, so
an unmodified ``PController("btle_rx")`` (snout/util/btle.py:53) can drive it.
"""
from __future__ import annotations

import sys

import click

from . import synth
from .scan import ArraySource, BtleScan, FileSource, ZigbeeScan

DEFAULTS = {"btle": dict(channels=(0, 39), default=37, timeout=10),      # snout/util/__init__.py:4-17
            "zigbee": dict(channels=(11, 26), default=11, timeout=10)}


def parse_channels(spec: str, proto: str):
    """'37', '37,38,39', '11:26' (inclusive range) — snout/util/iot_click.py:46-92."""
    lo, hi = DEFAULTS[proto]["channels"]
    out = []
    for part in str(spec).split(","):
        part = part.strip()
        if ":" in part:
            a, b = part.split(":")
            out.extend(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    for c in out:
        if not lo <= c <= hi:
            raise click.BadParameter(f"channel {c} outside {lo}..{hi} for {proto}")
    return out or [DEFAULTS[proto]["default"]]


@click.group()
def main():
    """MI355X-native receive path behind Snout's scan interface."""


FORMATS = {"cf32": 0, "sc8": 1, "sc16": 2}


def _source(proto, iq, synthetic, channels, seconds, fmt="cf32"):
    if iq:
        return FileSource(iq, FORMATS[fmt])
    if not synthetic:
        raise click.UsageError("give --iq FILE or --synthetic (no live SDR in this build)")
    n = int(seconds * 4e6)
    gen = synth.btle_capture if proto == "btle" else synth.zigbee_capture
    return ArraySource({ch: gen(n, channel=ch, seed=ch)[0] for ch in channels})


def _scan_options(f):
    for opt in reversed([
        click.option("-c", "--channels", default=None, help="e.g. 37 | 37,38 | 11:26"),
        click.option("-n", "--packets", type=int, default=None, help="stop after N packets"),
        click.option("-t", "--timeout", type=float, default=None, help="seconds of capture per channel"),
        click.option("-f", "--filename", default=None, help="dump file"),
        click.option("--iq", type=click.Path(exists=True), default=None, help="capture file"),
        click.option("--format", "fmt", type=click.Choice(sorted(FORMATS)), default="cf32",
                     help="sample format of --iq: cf32, sc8 (hackrf_transfer int8), sc16"),
        click.option("--synthetic", is_flag=True, help="generate a synthetic capture"),
        click.option("--seconds", type=float, default=0.25, help="length of a synthetic capture"),
    ]):
        f = opt(f)
    return f


@main.group()
def btle():
    """Bluetooth LE advertising channels."""


@btle.command("scan")
@_scan_options
@click.option("--summary", is_flag=True, help="print the device table at the end (snout/util/btle.py:202-240)")
def btle_scan(channels, packets, timeout, filename, iq, fmt, synthetic, seconds, summary):
    chs = parse_channels(channels or DEFAULTS["btle"]["default"], "btle")
    scan = BtleScan(channels=chs, source=_source("btle", iq, synthetic, chs, seconds, fmt),
                    timeout=timeout, packet_threshold=packets, filename=filename)
    scan.events.on("btle.packet-received",
                   lambda message: click.echo(message.raw.decode().rstrip("\n")))
    msgs = scan.run()
    if summary:
        from .devices import DeviceTable
        click.echo(DeviceTable().extend(msgs).render())
    click.echo(f"{len(msgs)} packets, {len({m.sender for m in msgs})} devices", err=True)


@main.group()
def zigbee():
    """IEEE 802.15.4 / Zigbee channels 11-26."""


@zigbee.command("scan")
@_scan_options
@click.option("--udp", is_flag=True, help="send RFtap datagrams to 127.0.0.1:52002 (scapy-radio)")
def zigbee_scan(channels, packets, timeout, filename, iq, fmt, synthetic, seconds, udp):
    chs = parse_channels(channels or DEFAULTS["zigbee"]["default"], "zigbee")
    scan = ZigbeeScan(channels=chs, source=_source("zigbee", iq, synthetic, chs, seconds, fmt),
                      timeout=timeout, packet_threshold=packets, udp=udp)
    def show(message):
        from .formats import parse_mhr
        try:
            h = parse_mhr(message.mpdu)
            hdr = f"{h['frame_type_name']} seq{h['seq']}"
        except ValueError:
            hdr = "short"
        click.echo(f"{message.timestamp:.6f} Ch{message.channel} LQI{message.lqi} {hdr} {message.mpdu.hex()}")

    scan.events.on("zigbee.packet-received", show)
    msgs = scan.run()
    if filename:
        from .formats import write_pcap
        write_pcap(filename, [(m.timestamp, m.mpdu) for m in msgs])
    click.echo(f"{len(msgs)} frames", err=True)


@main.command("btle-rx")
@click.option("-c", "channel", type=int, default=37)
@click.option("-g", "gain", type=int, default=6, help="accepted for btle_rx compatibility")
@click.option("-a", "access", default="8e89bed6")
@click.option("-k", "crcinit", default="555555")
@click.option("--iq", type=click.Path(exists=True), required=True)
@click.option("--format", "fmt", type=click.Choice(sorted(FORMATS)), default="cf32")
def btle_rx(channel, gain, access, crcinit, iq, fmt):
    """Drop-in for the `btle_rx` child: same argv (snout/util/btle.py:63-68), same stdout lines."""
    scan = BtleScan(channels=[channel], source=FileSource(iq, FORMATS[fmt]), timeout=None,
                    access_addr=int(access, 16), crc_init=int(crcinit, 16))
    for line in scan.lines(channel):
        sys.stdout.write(line.decode())
    sys.stdout.flush()


if __name__ == "__main__":
    main()
