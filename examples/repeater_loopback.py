#!/usr/bin/env python3
"""Full-duplex timed loop through driver=sx on an MI355X, the call pattern of the reference's
example/linear_repeater.py (read one block, process it, write it back with timeNs = RX time + latency),
run for a fixed number of blocks and checked instead of left running:

  * every TX block lands exactly `latency` samples after the RX block it answers (the reference's
    distinguishing feature, README.md:28-35), verified by reading the synthetic DAC-rate sink back;
  * RX timestamps advance by exactly one block.

  python3 examples/repeater_loopback.py [--rate 75000|300000] [--blocks 200] [--block 256] [--clock virtual|wall]

With a real SoapySDR installed and this repo's module built against it the same
script runs with `import SoapySDR` instead of the stand-in below."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sxxcvr_amd.soapy as SoapySDR          # noqa: E402  (stand-in for the SoapySDR Python module)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rate", type=float, default=75000.0)
    ap.add_argument("--blocks", type=int, default=200)
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--latency", type=int, default=8 * 256, help="RX-to-TX distance in samples")
    ap.add_argument("--clock", default="virtual", choices=["virtual", "wall"])
    args = ap.parse_args()

    dev = SoapySDR.Device({"driver": "sx", "clock": args.clock, "decim": "auto", "interp": "auto"})
    for d in (SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_TX):
        dev.setSampleRate(d, 0, args.rate)
    dev.setFrequency(SoapySDR.SOAPY_SDR_RX, 0, 432.55e6)
    dev.setFrequency(SoapySDR.SOAPY_SDR_TX, 0, 434.55e6)
    dev.setGain(SoapySDR.SOAPY_SDR_RX, 0, 55.0)
    dev.setGain(SoapySDR.SOAPY_SDR_TX, 0, 40.0)
    ratio = int(dev.readSetting("TX_INTERP"))
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, SoapySDR.SOAPY_SDR_CF32, [0], {"period": str(args.block)})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, SoapySDR.SOAPY_SDR_CF32, [0], {"threshold": "0", "period": str(args.block)})
    dev.activateStream(rx)
    dev.activateStream(tx)

    buf = np.zeros(args.block, dtype=np.complex64)
    dt = int(round(args.latency * 1e9 / args.rate))
    sent = []
    last_t = None
    for _ in range(args.blocks):
        r = dev.readStream(rx, [buf], len(buf))
        assert r.ret == len(buf) and r.flags & SoapySDR.SOAPY_SDR_HAS_TIME, r
        if last_t is not None:
            assert SoapySDR.timeNsToTicks(r.timeNs, args.rate) - SoapySDR.timeNsToTicks(last_t, args.rate) == len(buf)
        last_t = r.timeNs
        buf *= np.float32(0.5)                                         # "process"
        w = dev.writeStream(tx, [buf], len(buf), flags=SoapySDR.SOAPY_SDR_HAS_TIME, timeNs=r.timeNs + dt)
        assert w.ret == len(buf), w
        sent.append((SoapySDR.timeNsToTicks(r.timeNs, args.rate) + args.latency, buf.copy()))

    # the sink: a unit-gain interpolator, so the DAC-rate stream decimated by `ratio` at the filter's group
    # delay returns (almost exactly) what was written, at the position its timestamp named
    pos, block = sent[-1]
    delay = (32 * ratio - 1) // 2                                      # (ntaps - 1) / 2 DAC samples
    dac = dev.txCapture(pos * ratio, len(block) * ratio)
    k = np.arange(32, len(block) - 32)
    got = dac[k * ratio + delay] if (k[-1] * ratio + delay) < len(dac) else None
    err = float(np.max(np.abs(got - block[k]))) if got is not None else float("nan")
    print("blocks %d  rate %.0f S/s  ratio %d  last RX time %d ns  TX placed at sample %d  "
          "max |sink - written| = %.2e" % (args.blocks, args.rate, ratio, last_t, pos, err))
    assert err < 0.1, err
    dev.deactivateStream(rx)
    dev.deactivateStream(tx)
    dev.closeStream(rx)
    dev.closeStream(tx)


if __name__ == "__main__":
    main()
