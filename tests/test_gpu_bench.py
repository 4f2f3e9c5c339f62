"""bench.py as the driver runs it: the JSON line's contract, every --config, the self-spawned multi-rank path
(gloo stand-in on a 1-GPU box; nccl = RCCL when two GPUs are visible), and the self-check of its outputs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "2", "--settle", "4", "--log2-samples", "22", "--no-cpu-baseline",
         "--no-through-device"]


def run_bench(extra, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)                                   # bench.py must start its own ranks
    e.update(env or {})
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                         timeout=timeout, env=e)
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert len(lines) == 1, run.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("config", ["2", "3rx", "3tx", "5", "5h"])
def test_bench_line_every_config(config):
    line = run_bench(["--config", config] + SMALL)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "verified"):
        assert key in line, key
    assert line["verified"] is True and line["config"]["verified_outputs"] >= 4096
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 2
    assert line["config"]["bench_config"] == config and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 300.0 < r["shader_mhz"] < 2600.0
    assert line["value"] > 0 and r["kernel_ms"] > 0


def test_bench_config3_full_duplex_line():
    """BASELINE config 3 as ONE workload: /8 RX and x8 TX side by side on two streams, both verified against the
    oracle, a roofline entry per direction, and the timed readStream -> writeStream loop through the Device with the
    latency check on every block (example/linear_repeater.py:40-69; SoapySX.cpp:950, :1012)."""
    line = run_bench(["--config", "3", "--steps", "3", "--warmup", "2", "--settle", "4", "--log2-samples", "22",
                      "--no-cpu-baseline"])
    assert line["verified"] is True and line["config"]["bench_config"] == "3" and line["n_gpus"] == 1
    assert "full-duplex" in line["config"]["workload"] and line["config"]["verified_outputs"] >= 8192
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    for d in ("rx", "tx"):
        assert r[d]["span_ms_per_step"] > 0 and 0 < r[d]["frac"] < 1 and r["alone_kernel_ms"][d] > 0
    assert r["algorithmic_bytes_per_step"] == 2 * 9 * (1 << 22)
    t = line["timed_loop"]
    assert "error" not in t, t
    assert t["latency_check_passed"] is True
    for blk in (256, 1024, 4096):
        b = t["%d_sample_blocks" % blk]
        assert b["blocks_off_position"] == 0 and b["round_trip_us_median"] > 0 and b["latency_samples"] == 3 * blk
    assert t["256_sample_blocks"]["latency_ns"] == 2560000          # 768 samples at 300 kS/s


def test_bench_reports_device_and_cpu_figures():
    line = run_bench(["--steps", "3", "--warmup", "2", "--settle", "4", "--log2-samples", "22"], timeout=900)
    assert line["verified"] is True
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert c["conversion_only"]["one_thread_MS/s"] > 0 and c["conversion_only"]["all_threads_MS/s"] > 0
    t = line["through_device"]
    assert "error" not in t, t
    assert t["256_sample_calls"]["readStream_us_per_call"] > 0 and t["65536_sample_calls"]["readStream_out_MS/s"] > 0
    cc = t["c_caller"]                                   # tools/devloop.c, built by build(), run as a child process
    assert "error" not in cc, cc
    assert 0 < cc["256"]["readStream_us_per_call"] < 100 and 0 < cc["4096"]["writeStream_us_per_call"] < 1000
    r = line["roofline"]
    assert r["kernel_ms_first_20"] > 0 and "board" in r
    if "error" not in r["board"]:
        assert r["board"]["samples"] >= 10 and r["power_w"] > 100 and r["gfx_mhz_smi"] > 100


def test_bench_starts_its_own_ranks_gloo_stand_in():
    """--gpus 2 with no launcher environment: bench.py spawns the two ranks itself.  On a 1-GPU box the ranks
    share the GPU and the collectives run over gloo (SXFIR_DIST_BACKEND=gloo); the code path is the real one."""
    line = run_bench(["--gpus", "2"] + SMALL, env={"SXFIR_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["backend"] == "gloo"
    assert line["config"]["channels_per_gpu"] == 8
    g = line["gather"]
    assert "error" not in g, g
    assert g["root_holds_own_channels"] is True and g["gathered_shape"][0] == 16


def test_bench_eight_ranks_gloo_stand_in():
    """BASELINE config 4's full layout on the 1-GPU box: --gpus 8 self-spawned, eight ranks share the GPU, 64 channels,
    collectives over gloo.  Everything but the transport is the code the 8-GPU run executes: sharding, the serial gather
    and the pipelined (chunked, overlapped) gather, verification on every rank, one JSON line."""
    line = run_bench(["--gpus", "8"] + SMALL, env={"SXFIR_DIST_BACKEND": "gloo"}, timeout=1500)
    assert line["n_gpus"] == 8 and line["verified"] is True and line["scaling"] == "weak"
    assert line["config"]["rccl_ranks"] == 8 and line["config"]["channels_per_gpu"] == 8
    g = line["gather"]
    assert "error" not in g, g
    assert g["gathered_shape"][0] == 64 and g["root_holds_own_channels"] is True
    o = g["overlapped"]
    assert o["gathered_shape"][0] == 64 and o["root_holds_own_channels"] is True and o["chunks_per_step"] == 4
    assert g["overlapped_value"] > 0 and 0 < g["link_bound_frac"]


def test_bench_under_the_drivers_launcher_gloo_stand_in():
    """The driver's own N > 1 command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...; exactly one JSON line must come out (rank 0's).
    On a 1-GPU box the two ranks share the GPU and the collectives run over gloo."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    e = dict(os.environ, SXFIR_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified"] is True and line["scaling"] == "weak"
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["channels_per_gpu"] == 8
    assert line["gather"]["gathered_shape"][0] == 16


def test_bench_two_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    line = run_bench(["--gpus", "2"] + SMALL)
    assert line["n_gpus"] == 2 and line["verified"] is True and line["config"]["backend"] == "nccl"
    g = line["gather"]
    assert "error" not in g, g
    assert g["root_holds_own_channels"] is True and g["gathered_shape"][0] == 16


def test_bench_two_ranks_gather_through_the_c_abi():
    """--gather capi: the exchange step through sxfir_comm_gather (librccl directly), serial and in steady state."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    line = run_bench(["--gpus", "2", "--gather", "capi"] + SMALL)
    assert line["n_gpus"] == 2 and line["verified"] is True
    g = line["gather"]
    assert "error" not in g, g
    assert g["via"].startswith("C ABI") and g["root_holds_own_channels"] is True and g["gathered_shape"][0] == 16
    assert g["overlapped"]["root_holds_own_channels"] is True and g["overlapped_value"] > 0


def test_bench_refuses_the_c_abi_gather_over_the_gloo_stand_in():
    """--gather capi is RCCL itself: with ranks sharing a GPU (the gloo stand-in) it must refuse, not hang in ncclCommInitRank."""
    e = dict(os.environ, SXFIR_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gather", "capi"] + SMALL,
                         capture_output=True, text=True, timeout=300, env=e)
    assert run.returncode != 0 and "one GPU per rank" in (run.stdout + run.stderr)


def test_bench_refuses_mismatched_world():
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True,
                         text=True, timeout=300, env=e)
    assert run.returncode != 0 and "WORLD_SIZE" in (run.stdout + run.stderr)
