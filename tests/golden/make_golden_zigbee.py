#!/usr/bin/env python3
"""Generate the 802.15.4 golden fixture under tests/golden/ (no reference import needed):

zigbee_ch15_4msps.cf32     98 304 cf32 samples at 4 Msps: six frames built the way the reference
                           transmitter builds them (Zigbee_tx/top_block.py:59-71: preamble, SFD,
                           length, PSDU with FCS, O-QPSK half-sine), CFO up to 40 kHz, AWGN.
zigbee_ch15_expected.json  what the receive path reports for it at the default lane shape
                           (core 2048, warm-up 512): every record field, and the PSDUs sent.

The records pin the CURRENT definition of the lane / stitching / sink rules (oracle_zigbee.c): a
change of those rules shows up here first.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    import numpy as np
    from snout_amd import synth
    from oracle import oracle_py
    x, truth = synth.zigbee_capture(98304, channel=15, seed=15, mean_gap=6000.0, cfo_max_hz=40e3,
                                    n_packets=6, max_len=80)
    assert len(truth) == 6
    x.tofile(os.path.join(HERE, "zigbee_ch15_4msps.cf32"))
    pk = oracle_py.zigbee_segment(x, channel=15, first_sample_index=1000)
    good = [bytes(p["bytes"][:p["len"]]) for p in pk if p["crc_ok"]]
    assert sorted(good) == sorted(t.payload for t in truth)
    json.dump({
        "first_sample_index": 1000, "channel": 15, "core": 2048, "warmup": 512, "threshold": 10,
        "sent": [{"sample_index": t.sample_index, "psdu": t.payload.hex()} for t in truth],
        "records": [{"sample_index": int(p["sample_index"]), "channel": int(p["channel"]), "len": int(p["len"]),
                     "crc_ok": int(p["crc_ok"]), "lqi": int(p["lqi"]), "aux": int(p["aux"]),
                     "bytes": bytes(p["bytes"][:p["len"]]).hex()} for p in pk],
    }, open(os.path.join(HERE, "zigbee_ch15_expected.json"), "w"), indent=1)
    print("wrote", len(pk), "records,", len(truth), "frames sent")


if __name__ == "__main__":
    main()
