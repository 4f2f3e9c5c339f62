"""Through-the-Device throughput (readStream / writeStream incl. PCIe and launch overheads); API-parity
figure for LABBOOK.md, never the bench value.  DB_RATE=<rate of the reference's table>; DB_PIN=1: reads / writes of 32768 samples and
more use page-locked caller memory (the DMA path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sxxcvr_amd.soapy as SoapySDR
RATE = float(os.environ.get("DB_RATE", "600000"))          # any rate of the reference's table: 600e3 .. 25e3 (decim = interp = auto)
for blk in (256, 4096, 65536, 1 << 20):
    dev = SoapySDR.Device({"driver": "sx", "clock": "virtual", "decim": "auto", "interp": "auto"})
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, RATE)
    dev.setSampleRate(SoapySDR.SOAPY_SDR_TX, 0, RATE)
    ratio = int(dev.readSetting("RX_DECIM"))
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, "CF32", [0], {"period": str(min(blk, 65536))})
    tx = dev.setupStream(SoapySDR.SOAPY_SDR_TX, "CF32", [0], {"period": str(min(blk, 65536))})
    dev.activateStream(rx); dev.activateStream(tx)
    buf = np.zeros(blk, dtype=np.complex64)
    if os.environ.get("DB_PIN") and blk >= 32768:
        # page-locked caller memory (sxfir_host_register): large reads are DMA-copied straight into it, no staging hop, no host copy
        import sxxcvr_amd
        sxxcvr_amd.pin_array(buf)
    n = max(4, min(2000, (1 << 24) // blk))
    dev.readStream(rx, [buf], blk)
    t0 = time.perf_counter()
    for _ in range(n):
        r = dev.readStream(rx, [buf], blk)
        assert r.ret == blk
    dt_rx = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        r = dev.writeStream(tx, [buf], blk)
        assert r.ret == blk, r
    dt_tx = (time.perf_counter() - t0) / n
    print("rate %g (ratio %d) block %8d: readStream %.1f us/call = %.2f MS/s out (%.2f MS/s wideband in) | writeStream %.1f us/call = %.2f MS/s" % (
        RATE, ratio, blk, dt_rx * 1e6, blk / dt_rx / 1e6, ratio * blk / dt_rx / 1e6, dt_tx * 1e6, blk / dt_tx / 1e6))
