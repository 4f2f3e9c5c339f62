"""Stress of the zero-copy RX path (the decimator stores straight into page-locked caller memory): N times
{fresh stream, a short read, one large read into a registered buffer}, every output against the first run's.
Prints where mismatches fall (index, value, position inside the 256-output tile of the direct pass).

    python3 tools/stress_direct_rx.py [iterations=300] [large=38310] [small=400]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sxxcvr_amd
import sxxcvr_amd.soapy as SoapySDR

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
large = int(sys.argv[2]) if len(sys.argv) > 2 else 38310
small = int(sys.argv[3]) if len(sys.argv) > 3 else 400
pinned = sxxcvr_amd.pin_array(np.zeros(1 << 19, dtype=np.complex64))
ref, bad_runs, hist = None, 0, {}
for it in range(iters):
    dev = SoapySDR.Device({"driver": "sx", "clock": "virtual"})
    dev.setSampleRate(SoapySDR.SOAPY_SDR_RX, 0, 600000.0)
    rx = dev.setupStream(SoapySDR.SOAPY_SDR_RX, "CF32", [0], {"period": "65536"})
    dev.activateStream(rx)
    b0 = np.zeros(small, dtype=np.complex64)
    assert dev.readStream(rx, [b0], small).ret == small
    pinned[:large] = 0
    assert dev.readStream(rx, [pinned], large).ret == large
    got = pinned[:large].copy()
    direct = int(dev.readSetting("RX_DIRECT_SAMPLES"))
    dev.deactivateStream(rx); dev.closeStream(rx); del dev
    if ref is None:
        ref = got
        print("direct samples per large read:", direct)
        continue
    bad = np.nonzero(got.view(np.uint64) != ref.view(np.uint64))[0]
    if bad.size:
        bad_runs += 1
        first_direct = large - direct
        for i in bad[:8]:
            k = (int(i) - first_direct) % 256
            hist[k] = hist.get(k, 0) + 1
        print("iteration %d: %d outputs differ, first at %d (offset %d in the direct pass, %d in its tile): got %r want %r%s" % (
            it, bad.size, bad[0], bad[0] - first_direct, (bad[0] - first_direct) % 256, got[bad[0]], ref[bad[0]],
            "; later read agrees" if np.array_equal(pinned[:large].view(np.uint64), ref.view(np.uint64)) else ""))
print("%d of %d runs differ from the first; positions inside the tile: %s" % (bad_runs, iters - 1, dict(sorted(hist.items()))))
