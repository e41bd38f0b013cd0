"""`snout {btle,zigbee} scan` — the CLI surface of the reference (snout/cli.py:41-56,220-261) for the
receive path: same option names and grammar (-c/--channels, -a/--active, -n/--num, -t/--timeout), samples
taken from a recorded capture file (--iq) or a synthetic capture (--synthetic) instead of a live SDR.
`--wideband` / `--sharded` receive every channel of a whole-band capture at once (one or several GPUs).

Console entries (setup.py): `snout` = this group, as the reference's (setup.py:52-54); `btle_rx` prints
btle_rx-format lines for `-c 37 -g 6 -a 8e89bed6 -k 555555 --iq FILE`, so an unmodified
``PController("btle_rx")`` (snout/util/btle.py:53) can drive it.  In-tree: `bin/snout`, `bin/btle_rx`,
`python -m snout_amd`.
"""
from __future__ import annotations

import sys

import click

from . import synth
from .scan import ArraySource, BtleScan, FileSource, StreamSource, WidebandSource, ZigbeeScan

DEFAULTS = {"btle": dict(channels=(0, 39), default=37, timeout=10),      # snout/util/__init__.py:4-17
            "zigbee": dict(channels=(11, 26), default=11, timeout=10)}


def parse_channels(spec, proto: str):
    """The grammar of the reference's ``ChannelsOption`` + ``get_channels``
    (snout/util/iot_click.py:46-92, snout/cli.py:139-183): ``all`` -> every channel of the protocol;
    ``a-b`` / ``a:b`` -> the inclusive range; a Python literal list / tuple (``[11,12,13]``, ``37,38``)
    -> those channels as given (a one-element list is doubled, as upstream does); an int -> that
    channel; nothing -> the protocol's default channel (the reference prompts for it).  Channels
    outside the protocol's range are refused (the reference prints the range and exits)."""
    import ast
    lo, hi = DEFAULTS[proto]["channels"]
    value = None if spec is None else str(spec).strip()
    try:
        if not value:
            out = [DEFAULTS[proto]["default"]]
        elif value.lower() == "all":
            out = list(range(lo, hi + 1))
        elif "-" in value or ":" in value:
            a, b = value.split("-" if "-" in value else ":")
            out = list(range(int(a), int(b) + 1))
        else:
            v = ast.literal_eval(value)
            if isinstance(v, (list, tuple)):
                out = [int(c) for c in v]
                if len(out) == 1:
                    out = out * 2               # upstream: value.extend(value)
            elif isinstance(v, (int, float)):
                out = [int(v)]
            else:
                raise ValueError(value)
    except (ValueError, SyntaxError, TypeError):
        raise click.BadParameter(f"channels {spec!r}: int (11), inclusive range (11:26), list ([11,12,13]) or all")
    if not out or out[0] < lo or out[-1] > hi or any(not lo <= c <= hi for c in out):
        raise click.BadParameter(f"Channels for the {proto} protocol must be in the range of [{lo}, {hi}]")
    return out


def stop_conditions(proto: str, num, timeout):
    """snout/cli.py:246-250: at least one stop condition.  The reference prompts for both when neither
    -n nor -t is given ("Packet Threshold (leave empty to disable)", "Decode Timeout (positive int)",
    default snout/util/__init__.py:4-9); without a terminal the prompts' defaults apply."""
    if num is not None and num < 0:
        raise click.BadOptionUsage("num", "--num INT : int must be >= 0")
    if num is None and timeout is None:
        if sys.stdin is not None and sys.stdin.isatty():
            v = click.prompt("Packet Threshold (leave empty to disable)", default="", show_default=False)
            num = int(v) if str(v).strip() else None
            timeout = float(click.prompt("Decode Timeout (positive int)", type=click.INT,
                                         default=DEFAULTS[proto]["timeout"]))
        else:
            timeout = float(DEFAULTS[proto]["timeout"])
    return (num or None), timeout


@click.group()
def main():
    """MI355X-native receive path behind Snout's scan interface."""


FORMATS = {"cf32": 0, "sc8": 1, "sc16": 2}


def _source(proto, iq, synthetic, channels, seconds, fmt="cf32", wideband=False, sharded=False, segment=1 << 24, batch=4):
    if wideband or sharded:
        pid = 0 if proto == "btle" else 1
        if iq:
            import os
            import stat
            try:
                regular = iq != "-" and stat.S_ISREG(os.stat(iq).st_mode)
            except OSError as e:
                raise click.UsageError(f"--iq {iq}: {e.strerror}")
            if not regular:
                raise click.UsageError("--wideband / --sharded read a capture FILE (live streams -- '-', a FIFO, a device -- "
                                       "are implemented for the single-channel scans only)")
            return WidebandSource(iq, pid, FORMATS[fmt], segment=segment, sharded=sharded, batch=batch)
        if not synthetic:
            raise click.UsageError("give --iq FILE or --synthetic (no live SDR in this build)")
        M = 40 if pid == 0 else 16
        x, _ = synth.wideband_capture(pid, int(seconds * M * 2e6) // M * M, seed=3 + pid)
        if fmt != "cf32":
            x = synth.quantize(x, FORMATS[fmt]).reshape(-1, 2)
        return WidebandSource(x, pid, FORMATS[fmt], segment=segment, sharded=sharded, batch=batch)
    if iq:
        return _file_or_stream(iq, FORMATS[fmt])
    if not synthetic:
        raise click.UsageError("give --iq FILE or --synthetic (no live SDR in this build)")
    n = int(seconds * 4e6)
    gen = synth.btle_capture if proto == "btle" else synth.zigbee_capture
    return ArraySource({ch: gen(n, channel=ch, seed=ch)[0] for ch in channels})


def _file_or_stream(iq: str, fmt: int):
    """``-`` (stdin), a FIFO or a character device is a live stream (`hackrf_transfer -r - | ... --iq -`): read once, in
    overlapping segments, -t by the wall clock; a regular file is a capture: -t is capture time."""
    import os
    import stat
    if iq == "-":
        return StreamSource("-", fmt)
    try:
        mode = os.stat(iq).st_mode
    except OSError as e:
        raise click.UsageError(f"--iq {iq}: {e.strerror}")
    if stat.S_ISFIFO(mode) or stat.S_ISCHR(mode) or stat.S_ISSOCK(mode):
        return StreamSource(iq, fmt)
    return FileSource(iq, fmt)


def _iq_path(ctx, param, value):
    if value is None or value == "-":
        return value
    import os
    if not os.path.exists(value):
        raise click.BadParameter(f"{value!r} does not exist")
    return value


def _scan_options(f):
    for opt in reversed([
        click.option("-c", "--channels", default=None,
                     help="Specify the channels. Int (11), inclusive range (11:26), list ([11,12,13]), or all (all)"),
        click.option("-a", "--active", is_flag=True, default=None, help="Active Scan (accepted; the receive path is passive)"),
        click.option("-n", "--num", "packets", type=int, default=None, help="Number of packets to scan for each channel"),
        click.option("-t", "--timeout", type=float, default=None,
                     help="Add a timeout restriction (in seconds) for each channel."),
        click.option("--wideband", is_flag=True,
                     help="--iq holds the whole band (BTLE: 80 Msps centred 2442 MHz, Zigbee: 32 Msps): every "
                          "channel is received at once through the polyphase channelizer"),
        click.option("--sharded", is_flag=True,
                     help="wideband: cut the capture into overlapping segments, one GPU per rank "
                          "(run under torch.distributed.run), records gathered on rank 0"),
        click.option("--segment", type=int, default=1 << 24, help="wideband: input samples per segment"),
        click.option("--batch", type=click.IntRange(1, 64), default=4,
                     help="wideband: segments handed to the GPU as one submission"),
        click.option("-f", "--filename", default=None, help="dump file"),
        click.option("--iq", callback=_iq_path, default=None,
                     help="capture file; '-' (stdin) or a FIFO: a live stream, read in overlapping segments, -t by the wall clock"),
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
def btle_scan(channels, active, packets, timeout, wideband, sharded, segment, batch, filename, iq, fmt, synthetic,
              seconds, summary):
    chs = parse_channels(channels, "btle")
    packets, timeout = stop_conditions("btle", packets, timeout)
    scan = BtleScan(channels=chs, source=_source("btle", iq, synthetic, chs, seconds, fmt, wideband, sharded, segment, batch),
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
@click.option("--lane-core", type=int, default=0,
              help="clock-recovery lane length in channel samples (multiple of 64; 0: the default shape, 6144 with warm-up 3072 for --wideband scans, 1024 for single-channel ones). The "
                   "reference's receiver is ONE sequential loop: a lane at least as long as the capture is that loop (DESIGN.md 6-3)")
@click.option("--lane-warmup", type=int, default=0, help="samples a lane's timing loop starts before its core (multiple of 64; 0 with --lane-core 0: the default shape; "
                   "0 with a --lane-core: 512)")
def zigbee_scan(channels, active, packets, timeout, wideband, sharded, segment, batch, filename, iq, fmt, synthetic,
                seconds, udp, lane_core, lane_warmup):
    chs = parse_channels(channels, "zigbee")
    packets, timeout = stop_conditions("zigbee", packets, timeout)
    src = _source("zigbee", iq, synthetic, chs, seconds, fmt, wideband, sharded, segment, batch)
    if lane_core or lane_warmup:
        src.rx_kw = dict(zb_core=lane_core, zb_warmup=lane_warmup)
    scan = ZigbeeScan(channels=chs, source=src, timeout=timeout, packet_threshold=packets, udp=udp)
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
@click.option("--iq", callback=_iq_path, required=True, help="capture file, or '-' / a FIFO for a live stream")
@click.option("--format", "fmt", type=click.Choice(sorted(FORMATS)), default="cf32")
def btle_rx(channel, gain, access, crcinit, iq, fmt):
    """Drop-in for the `btle_rx` child: same argv (snout/util/btle.py:63-68), same stdout lines.
    `hackrf_transfer -r - ... | btle_rx -c 37 --format sc8 --iq -` receives a live stream: lines appear segment by segment."""
    scan = BtleScan(channels=[channel], source=_file_or_stream(iq, FORMATS[fmt]), timeout=None,
                    access_addr=int(access, 16), crc_init=int(crcinit, 16))
    for line in scan.lines(channel):
        sys.stdout.write(line.decode())
        sys.stdout.flush()


def btle_rx_main():
    """Console entry `btle_rx`: the name the reference resolves on $PATH (snout/util/btle.py:53)."""
    btle_rx.main(prog_name="btle_rx")


if __name__ == "__main__":
    main()
