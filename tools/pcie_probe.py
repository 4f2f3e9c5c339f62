"""What the box's PCIe link gives a device->host / host->device copy of pinned memory (the ceiling of the
readStream / writeStream figures bench.py reports as through_device), by transfer size."""
import time
import torch

dev = torch.device("cuda", 0)
for log2 in (16, 20, 23, 26, 28):
    n = 1 << log2
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = max(4, min(2000, (1 << 31) // n))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%s %9d bytes: %8.1f us  %6.2f GB/s" % (name, n, dt * 1e6, n / dt / 1e9))
# two copies in flight on two streams (both SDMA engines)
n = 1 << 26
d = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
s = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(16):
    for k in range(2):
        with torch.cuda.stream(s[k]):
            h[k].copy_(d[k], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 32
print("D2H two streams, 64 MiB each: %.2f GB/s aggregate" % (n / dt / 1e9))
