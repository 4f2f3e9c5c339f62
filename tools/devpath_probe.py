"""Per-call times of large readStream / writeStream calls through the Device (pageable and page-locked caller
memory): shows whether a figure of bench.py's through_device block is steady or carried by outliers."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sxxcvr_amd
import sxxcvr_amd.soapy as SoapySDR

blk = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for pin in (False, True):
    dev = SoapySDR.Device({"driver": "sx", "clock": "virtual"})
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 600000.0)
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, "CF32", [0], {"period": "65536"})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, "CF32", [0], {"period": "65536"})
    dev.activateStream(rx)
    dev.activateStream(tx)
    buf = np.zeros(blk, dtype=np.complex64)
    if pin:
        sxxcvr_amd.pin_array(buf)
    for name, fn in (("readStream", lambda: dev.readStream(rx, [buf], blk)), ("writeStream", lambda: dev.writeStream(tx, [buf], blk))):
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            r = fn()
            ts.append((time.perf_counter() - t0) * 1e6)
            assert r.ret == blk
        a = np.array(ts[2:])
        print("%s %s %d samples: us per call median %.0f min %.0f max %.0f | %s" % (
            "page-locked" if pin else "pageable", name, blk, np.median(a), a.min(), a.max(), " ".join("%.0f" % t for t in ts)))
    if pin:
        sxxcvr_amd.unpin_array(buf)
