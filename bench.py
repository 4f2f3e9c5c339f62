#!/usr/bin/env python3
"""Benchmark of the MI355X resampling hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one streaming pass of the 128-tap decimate-by-4 CF32 polyphase
FIR (sxfir_decimate through the C ABI: one kernel launch, history carry-over fused)
over the rank's resident synthetic IQ block (2^28 complex samples per GPU,
already in HBM when the timed region starts); consecutive steps are consecutive
blocks of one continuous stream (filter history carried over on the GPU).  N = 1 runs BASELINE config 2
(1 channel); N > 1 runs config 4's layout (8 independent channels per GPU,
8*N in total, no data-path collective), and measures the RCCL gather of the
decimated output to rank 0 separately (reported under "gather", never part of
`value`: it is xGMI-link bound).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NTAPS, DECIM = 128, 4
SEED = 0x51255
BYTES_PER_INPUT_SAMPLE = 8 + 8 / DECIM          # 8 B read + 2 B written
FLOP_PER_INPUT_SAMPLE = 4 * NTAPS / DECIM
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0                            # measured float4 copy ceiling (same guide)


def cpu_baseline(seconds_target=12.0):
    """The build's own CPU FIR (the reference has none), order-matched fp32
    oracle, timed on this host on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    oracle_lib.build()
    cpuinfo = open("/proc/cpuinfo").read()
    fast = (" avx2" in cpuinfo) and (" fma" in cpuinfo)
    orc = oracle_lib.Oracle(fast=fast)
    taps = orc.design_lowpass(NTAPS, DECIM)
    threads = orc.max_threads()
    model = "unknown"
    for line in cpuinfo.splitlines():
        if line.startswith("model name"):
            model = line.split(":", 1)[1].strip()
            break
    n_probe = 1 << 21
    x = orc.synth_iq(SEED, 0, 0, n_probe)
    t0 = time.perf_counter()
    orc.decim_f32(taps, DECIM, x, 2, 4, threads=1)
    t1 = time.perf_counter() - t0
    one_thread = n_probe / t1 / 1e6
    # bounded sample: the first 2^26 input samples of the workload's stream, filtered repeatedly
    # until about `seconds_target` of wall time has been spent
    n = 1 << 26
    x = orc.synth_iq(SEED, 0, 0, n)
    orc.decim_f32(taps, DECIM, x, 2, 4, threads=threads)          # warm-up (page faults, thread pool)
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < seconds_target and reps < 1000:
        orc.decim_f32(taps, DECIM, x, 2, 4, threads=threads)
        reps += 1
        dt = time.perf_counter() - t0
    return {
        "value": round(n * reps / dt / 1e6, 2),
        "unit": "MS/s (complex input samples)",
        "cores": threads,
        "kind": "port",
        "sample": "first %d input samples of the same synthetic channel-0 stream filtered %d times, %d threads "
                  "(OpenMP over output blocks), %s build; the reference has no software FIR, this is the "
                  "build's own CPU FIR" % (n, reps, threads, "AVX2+FMA" if fast else "portable"),
        "one_thread_value": round(one_thread, 2),
        "cpu_model": model,
        "seconds": round(dt, 2),
    }


def measure_gather(world, y, total_channels, host_collectives, cdev, per_gpu, elapsed, steps):
    """Exchange step of BASELINE config 4: RCCL gather of every rank's decimated output to rank 0 over xGMI,
    timed after (and outside) the timed region."""
    import torch
    import torch.distributed as dist
    import sxxcvr_amd.dist as sxdist
    gather = None
    if world > 1:
        # exchange step of config 4: decimated output of every rank to rank 0 over xGMI
        yg = y.cpu() if host_collectives else y
        for _ in range(2):
            sxdist.gather_channels(yg, total_channels, dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            sxdist.gather_channels(yg, total_channels, dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        g = (time.perf_counter() - g0) / reps
        t = torch.tensor([g], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g = float(t.item())
        peer_bytes = y.numel() * 8
        gather = {
            "ms": round(g * 1e3, 3),
            "bytes_per_peer": peer_bytes,
            "GB/s_into_root": round(peer_bytes * (world - 1) / g / 1e9, 2),
            "GB/s_per_link": round(peer_bytes / g / 1e9, 2),
            "value_with_gather": round(world * per_gpu * 1.0 / (elapsed / steps + g) / 1e6, 1),
            "note": "gather of the decimated output is xGMI per-link bound (~153 GB/s per peer) and not part "
                    "of value",
        }

    return gather


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip's power management needs ~20 launches (~10 ms) of this kernel to settle
    # (kernel time swings 0.50 -> 0.77 -> 0.60 ms before it does), so warm up past that
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--log2-samples", type=int, default=28, help="input samples per GPU (log2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle", type=int, default=150, help="untimed launches before the warm-up steps (clock settling)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import sxxcvr_amd
    from sxxcvr_amd import dist as sxdist
    from sxxcvr_amd.resampler import DECIMATE

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # SXFIR_DIST_BACKEND=gloo is a dry-run aid: it lets N ranks share the GPUs that exist (rank -> GPU
    # local_rank % device_count) and moves the collectives to the host, to exercise this code path on a
    # 1-GPU box.  The real multi-GPU run uses the default: nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("SXFIR_DIST_BACKEND")
    host_collectives = backend == "gloo"
    rank, local_rank, world = sxdist.env_rank()
    gpu_index = local_rank % torch.cuda.device_count() if host_collectives else local_rank
    torch.cuda.set_device(gpu_index)
    rank, local_rank, world = sxdist.init_process_group(backend=backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", gpu_index)
    cdev = torch.device("cpu") if host_collectives else dev

    per_gpu = 1 << args.log2_samples
    if world == 1:
        nchan_local, total_channels = 1, 1
        workload = "1xMI355X: 128-tap polyphase decim-by-4, 1 ch CF32 streaming (BASELINE config 2)"
    else:
        nchan_local, total_channels = 8, 8 * world
        workload = ("%dxMI355X: %d independent CF32 channels sharded 8/GPU, 128-tap decim-by-4 "
                    "(BASELINE config 4 layout)" % (world, total_channels))
    n_in = per_gpu // nchan_local
    lo, hi = sxdist.shard_channels(total_channels, world, rank)
    assert hi - lo == nchan_local

    taps = sxxcvr_amd.design_lowpass(NTAPS, DECIM)
    plan = sxxcvr_amd.Resampler(DECIMATE, taps, DECIM, nchan=nchan_local, device=gpu_index)
    x = torch.empty((nchan_local, n_in), dtype=torch.complex64, device=dev)
    sxxcvr_amd.synth_fill(x, SEED, first_channel=lo, start=0)
    y = torch.empty((nchan_local, n_in // DECIM), dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()

    def step():
        # one streaming pass: the block is the next 2^28 samples of a continuous stream (the filter
        # history carries over from the previous step, as it does in readStream)
        plan.process(x, out=y)

    # Setup, untimed and independent of --warmup: let the chip's power management settle on this kernel (its
    # time swings 0.49 -> 0.84 -> 0.60 ms over the first ~20 launches); same launches
    # as a step, then the stream restarts at position 0.
    if args.settle > 0:
        plan.time_decimate_ptr(x.data_ptr(), n_in, x.stride(0) if nchan_local > 1 else n_in, y.data_ptr(),
                               y.stride(0) if nchan_local > 1 else n_in // DECIM, args.settle,
                               torch.cuda.current_stream(dev).cuda_stream)
        plan.reset()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel alone, HIP events on the launch stream (not torch's event API)
    stream = torch.cuda.current_stream(dev).cuda_stream
    plan.reset()
    torch.cuda.synchronize()
    kernel_ms = plan.time_decimate_ptr(x.data_ptr(), n_in, x.stride(0) if nchan_local > 1 else n_in, y.data_ptr(),
                                       y.stride(0) if nchan_local > 1 else n_in // DECIM, min(max(args.steps, 20), 200), stream)
    achieved = BYTES_PER_INPUT_SAMPLE * per_gpu / (kernel_ms * 1e-3) / 1e9

    def emit(gather):
        """Rank 0 prints the one JSON line."""
        if rank == 0:
            ms_per_step = elapsed / args.steps * 1e3
            value = world * per_gpu * args.steps / elapsed / 1e6
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp) and world == 1 and args.log2_samples == 28:     # measured for exactly this launch
                try:
                    traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            line = {
                "metric": "complex MS/s, 128-tap decim-by-4 CF32 (input rate, whole job)",
                "value": round(value, 1),
                "unit": "MS/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f32",
                "data": "synthetic",
                "config": {
                    "workload": workload,
                    "ntaps": NTAPS, "decim": DECIM, "format": "CF32",
                    "channels_per_gpu": nchan_local, "input_samples_per_gpu": per_gpu,
                    "untimed_settle_launches": args.settle,
                    "output_MS/s": round(value / DECIM, 1),
                    "per_gpu_MS/s": round(value / world, 1),
                },
                "roofline": {
                    "bound": "hbm",
                    "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "kernel": "sxfir::decim4_tile_kernel<128>",
                    "kernel_ms": round(kernel_ms, 4),
                    "algorithmic_bytes_per_launch": int(BYTES_PER_INPUT_SAMPLE * per_gpu),
                    "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_GBS, 4),
                    "fp32_TFLOPs": round(FLOP_PER_INPUT_SAMPLE * per_gpu / (kernel_ms * 1e-3) / 1e12, 2),
                },
            }
            if gather is not None:
                line["gather"] = gather
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline()
            print(json.dumps(line), flush=True)

    # The exchange step (config 4's gather over xGMI) is reported beside `value`, never inside it, and must
    # not be able to take the line down with it: errors are recorded, and a watchdog prints the line without
    # the gather figures if the collective does not come back.
    gather = None
    if world > 1:
        import threading
        finished = threading.Event()

        def watchdog():
            if not finished.wait(240.0):
                emit({"error": "gather did not complete within 240 s"})
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            gather = measure_gather(world, y, total_channels, host_collectives, cdev, per_gpu, elapsed, args.steps)
        except Exception as e:
            gather = {"error": "%s: %s" % (type(e).__name__, e)}
        finished.set()
    emit(gather)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
