import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sxxcvr_amd.soapy as S
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
N = int(os.environ.get("SOAK_N", "40"))
for i in range(N):
    dev = S.Device({"driver": "sx", "clock": "virtual", "channels": "2", "decim": "auto", "interp": "auto"})
    dev.setSampleRate(S.SOAPY_SDR_RX, 0, 300000.0 if i % 2 else 75000.0)
    rx = dev.setupStream(S.SOAPY_SDR_RX, S.SOAPY_SDR_CF32, [], {})
    tx = dev.setupStream(S.SOAPY_SDR_TX, S.SOAPY_SDR_CF32, [], {})
    dev.activateStream(rx); dev.activateStream(tx)
    bufs = [np.zeros(1000, np.complex64) for _ in range(2)]
    for _ in range(5):
        assert dev.readStream(rx, bufs, 1000).ret == 1000
        assert dev.writeStream(tx, bufs, 1000).ret == 1000
    dev.deactivateStream(rx); dev.deactivateStream(tx)
    dev.closeStream(rx); dev.closeStream(tx)
    del dev
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(str(N) + " devices in %.1f s; free HBM before/after: %.1f / %.1f MiB" % (time.time() - t0, free0 / 2**20, free1 / 2**20))
