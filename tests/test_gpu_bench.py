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
         "--no-through-device", "--no-rate-table", "--alone-seconds", "0.05"]


def check_multi_rank_fields(line, world, backend):
    """What makes an N > 1 line self-contained: the communicator's own rank count, the GPUs' PCI
    addresses, every rank's kernel alone and together with the in-job efficiency, and a gather that rank 0 has checked
    block by block (checksums of every channel + oracle windows on the first channel of every rank's block)."""
    c = line["config"]
    assert c["rccl_ranks"] == world and c["backend"] == backend
    assert len(c["gpus"]) == world and all(isinstance(g, str) and g.count(":") == 2 for g in c["gpus"]), c["gpus"]
    assert c["distinct_gpus"] == (world if backend == "nccl" else len(set(c["gpus"])))
    assert "HSA_ENABLE_IPC_MODE_LEGACY" in c["env"] and c["control_group"] == "gloo"
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == list(range(world))
    for r in pr:
        assert r["alone_ms"] > 0 and r["together_ms"] > 0 and r["gpu"] == c["gpus"][r["rank"]]
        assert r["alone_launches"] >= 50 and r["together_launches"] >= 50
        assert "power_w_alone" in r and "power_w_together" in r
        assert r["timed_kernel_ms"] > 0 and 0 < r["timed_frac"] < 1
    mean = lambda k: sum(r[k] for r in pr) / world
    assert abs(line["efficiency_kernel_only"] - mean("alone_ms") / mean("together_ms")) < 5e-3 * line["efficiency_kernel_only"]
    e = line["efficiency_per_rank"]
    assert 0 < e["min"] <= e["median"] <= e["max"]
    f = line["roofline"]["frac_per_gpu"]
    assert 0 < f["min"] <= f["median"] <= f["max"] < 1 and f["min"] == min(r["timed_frac"] for r in pr)
    g = line["gather"]
    assert "error" not in g, g
    assert g["rccl_ranks"] == world
    assert line["gather_verified"] is True
    checks = [g["check"]] + ([g["overlapped"]["check"]] if "overlapped" in g and "check" in g["overlapped"] else [])
    for k in checks:
        assert k["verified"] is True and k["bad_channels"] == [] and k["oracle_ok"] is True
        assert k["peer_blocks_checked"] == world - 1 and k["channels_checksummed"] == 8 * world
        assert k["oracle_outputs_on_first_channel_of_every_block"] >= 2048 * world and k["stream_block"] >= 1
    return checks


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


@pytest.mark.parametrize("config", ["2", "3rx", "3tx", "5", "5h", "rx16", "rx48", "rx96", "tx4", "tx16", "tx32", "tx48", "tx96"])
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
    # the regimes of the same kernel, as roofline fractions in the one record
    bytes_ = r["algorithmic_bytes_per_launch"]
    # (the line rounds the times to 0.1 us; at this test's size a launch is ~10 us)
    assert abs(r["frac_cold_first_20"] - bytes_ / (r["kernel_ms_first_20"] * 1e-3) / 8e12) < 0.02 * r["frac_cold_first_20"]
    assert abs(r["frac_back_to_back_after_idle"] - bytes_ / (r["kernel_ms_back_to_back_loop"] * 1e-3) / 8e12) \
        < 0.02 * r["frac_back_to_back_after_idle"]
    assert r["frac_while_sampled"] is None or r["frac_while_sampled"] > 0
    assert line["config"]["gpus"] and line["config"]["distinct_gpus"] == 1 and line["config"]["gpu_arch"].startswith("gfx950")


def test_bench_line_carries_the_table_of_the_references_rates():
    """The default line (config 2) reports every rate of the reference's table beside the value: twelve rows, each checked
    against the oracle before it is timed."""
    line = run_bench([a for a in SMALL if a != "--no-rate-table"])
    t = line["rate_table"]
    assert "error" not in t and t["verified"] is True, t
    assert sorted(t["rows"]) == sorted(d + str(r) for d in ("rx", "tx") for r in (4, 8, 16, 32, 48, 96))
    for name, row in t["rows"].items():
        assert row["verified"] is True and row["outputs_compared"] >= 2048 and row["ms_per_2^28"] > 0, (name, row)
    # the two slowest rates run LDS-tiled kernels like the others (they ran the generic kernels, 80 x and 15 x slower, until
    # round 5; at this test's block size a /96 call is 85 tiles on 512 workgroup slots, so the bound is loose)
    # (round 6: (tile, block) items for small calls -- the same call is now within 2.5 x of /32's time per sample, 1.5 x on TX)
    assert t["rows"]["rx96"]["ms_per_2^28"] < 2.5 * t["rows"]["rx32"]["ms_per_2^28"]
    assert t["rows"]["tx96"]["ms_per_2^28"] < 1.5 * t["rows"]["tx32"]["ms_per_2^28"]
    # ... and the size curve beside it: kernel time per call at the call sizes the API issues, with the launch geometry
    c = line["size_curve"]
    assert "error" not in c, c
    assert sorted(c["rows"]) == sorted(t["rows"])
    for name, by_size in c["rows"].items():
        assert sorted(by_size) == ["2^18", "2^20", "2^22", "2^24"], (name, sorted(by_size))
        for sz, r in by_size.items():
            assert r["us_per_call"] > 0 and r["workgroups"] >= 1 and r["slots"] >= 256 and r["x_of_ratio_32"] > 0, (name, sz, r)
    # the slowest rates deal (tile, block) / (tile, phase block) items at these sizes ...
    assert c["rows"]["rx96"]["2^22"]["items_per_tile"] == 6 and c["rows"]["rx48"]["2^22"]["items_per_tile"] == 3
    assert c["rows"]["tx96"]["2^22"]["items_per_tile"] == 6 and c["rows"]["tx32"]["2^22"]["items_per_tile"] == 2
    # ... and are within 2 x of /32 and x32 per sample from 2^22 samples up (round 5: 4.7 x and 2.0 x at 2^22)
    for sz in ("2^22", "2^24"):
        assert c["rows"]["rx96"][sz]["x_of_ratio_32"] < 2.0 and c["rows"]["rx48"][sz]["x_of_ratio_32"] < 2.0, c["rows"]["rx96"]
        assert c["rows"]["tx96"][sz]["x_of_ratio_32"] < 1.6 and c["rows"]["tx48"][sz]["x_of_ratio_32"] < 1.6, c["rows"]["tx96"]


def test_bench_times_the_kernel_of_non_symmetric_taps():
    """--asymmetric-taps: config 2's shape with a filter that is not bit-symmetric runs decim4_tile_kernel<128>; its own row,
    verified against the oracle with the same taps."""
    line = run_bench(["--asymmetric-taps"] + SMALL)
    assert line["verified"] is True and line["config"]["bench_config"] == "2-asymmetric-taps"
    assert "decim4_tile_kernel<128>" in line["roofline"]["kernel"] and "NON-symmetric" in line["config"]["workload"]
    assert line["roofline"]["traffic"] is None


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
    assert abs(r["duplex_vs_serial"] - line["ms_per_step"] / r["sum_alone_ms"]) < 0.05 * r["duplex_vs_serial"]
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
    checks = check_multi_rank_fields(line, 2, "gloo")
    assert len(checks) == 2                                   # the serial gather and the pipelined one
    assert checks[0]["stream_block"] != checks[1]["stream_block"]      # each certified gather carries a block of its own


@pytest.mark.parametrize("who", ["1", "0"])
def test_bench_fails_on_a_corrupted_peer_block(who):
    """Test hook SXFIR_BENCH_CORRUPT_RANK: that rank flips one bit of what it sends after stating its checksums.  The line
    must come out with verified false, name the channel, and the job must exit non-zero (rank 1: a peer block that
    travelled; rank 0: the root's own block)."""
    e = dict(os.environ, SXFIR_DIST_BACKEND="gloo", SXFIR_BENCH_CORRUPT_RANK=who)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True,
                         text=True, timeout=600, env=e)
    assert run.returncode != 0, run.stdout[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:] + run.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["verified"] is False and line["gather_verified"] is False
    k = line["gather"]["check"]
    assert k["verified"] is False and len(k["bad_channels"]) == 1
    assert 8 * int(who) <= k["bad_channels"][0] < 8 * int(who) + 8
    assert line["gather"]["overlapped"]["check"]["verified"] is False


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
    assert len(check_multi_rank_fields(line, 8, "gloo")) == 2


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
    check_multi_rank_fields(line, 2, "gloo")


def test_bench_two_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    line = run_bench(["--gpus", "2"] + SMALL)
    assert line["n_gpus"] == 2 and line["verified"] is True and line["config"]["backend"] == "nccl"
    g = line["gather"]
    assert "error" not in g, g
    assert g["root_holds_own_channels"] is True and g["gathered_shape"][0] == 16
    assert len(check_multi_rank_fields(line, 2, "nccl")) == 2


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
    assert len(check_multi_rank_fields(line, 2, "nccl")) == 2


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
