"""The N > 1 layout on CPU: channel sharding and the gather of decimated
outputs, world_size 2, gloo backend (the GPU path uses the same code with
backend nccl = RCCL).  The payload here is produced by the oracle, standing in
for what each rank's GPU decimator would hold."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sxxcvr_amd import dist as sxdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_channels_partitions_exactly():
    for n in (1, 7, 8, 63, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [sxdist.shard_channels(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sxdist.shard_channels(64, 8, 3) == (24, 32)                 # BASELINE config 4: 8 channels per GPU
    with pytest.raises(ValueError):
        sxdist.shard_channels(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_channels, n_out, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        r, lr, w = sxdist.init_process_group(backend="gloo")
        assert (r, w) == (rank, world)
        orc = oracle_lib.Oracle()
        taps = orc.design_lowpass(128, 4)
        lo, hi = sxdist.shard_channels(n_channels, world, rank)
        local = np.stack([orc.decim_f32(taps, 4, orc.synth_iq(0x51255, c, 0, 4 * n_out), 2, 4) for c in range(lo, hi)])
        out = sxdist.gather_channels(torch.from_numpy(local), n_channels, dst=0)
        if rank == 0:
            q.put(("ok", out.numpy()))
        else:
            assert out is None
            q.put(("ok", None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put(("err", repr(e)))


@pytest.mark.parametrize("n_channels", [16, 5])
def test_gather_two_ranks_gloo(oracle, n_channels):
    world, n_out = 2, 300
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_channels, n_out, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(s == "ok" for s, _ in results), results
    gathered = [a for _, a in results if a is not None][0]
    taps = oracle.design_lowpass(128, 4)
    want = np.stack([oracle.decim_f32(taps, 4, oracle.synth_iq(0x51255, c, 0, 4 * n_out), 2, 4)
                     for c in range(n_channels)])
    assert gathered.shape == want.shape
    assert np.array_equal(gathered.view(np.uint64), want.view(np.uint64))


def _cert_worker(rank, world, port, n_channels, n, corrupt_rank, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        sxdist.init_process_group(backend="gloo")
        lo, hi = sxdist.shard_channels(n_channels, world, rank)
        g = torch.Generator().manual_seed(1000 + rank)
        local = torch.complex(torch.randn(hi - lo, n, generator=g), torch.randn(hi - lo, n, generator=g))
        sums = sxdist.block_checksums(local)                 # stated from what the sender holds ...
        if rank == corrupt_rank:
            local.view(torch.int64)[1, n // 2] ^= 1 << 33     # ... then one bit flips on the way
        stated = sxdist.exchange_checksums(sums, n_channels)
        full = sxdist.gather_channels(local, n_channels, dst=0)
        bad = sxdist.check_gathered(full, stated) if rank == 0 else None
        # a block that landed in another rank's place is caught too
        swapped = None
        if rank == 0 and corrupt_rank < 0:
            per = n_channels // world
            perm = full.clone()
            perm[:per], perm[per:2 * per] = full[per:2 * per], full[:per]
            swapped = sxdist.check_gathered(perm, stated)
        q.put(("ok", bad, swapped, tuple(stated.shape)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc(), None, None))


@pytest.mark.parametrize("corrupt_rank", [-1, 1, 0])
def test_gather_is_certified_by_sender_checksums_gloo(corrupt_rank):
    """What bench.py's N > 1 line relies on (GatherCertifier): every rank states two 64-bit checksums per channel of the
    block it sends, the tables are all-gathered, the root recomputes them over the gathered tensor.  A clean gather passes;
    one flipped bit in a peer's block (or the root's own) names exactly that channel; two blocks in each other's place
    fail on every channel of both."""
    world, n_channels, n = 2, 16, 4097
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cert_worker, args=(r, world, port, n_channels, n, corrupt_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] == "ok" for r in results), results
    root = [r for r in results if r[1] is not None][0]
    assert root[3] == (n_channels, 2)
    if corrupt_rank < 0:
        assert root[1] == [] and root[2] == list(range(16))
    else:
        assert root[1] == [8 * corrupt_rank + 1]


def test_block_checksums_definition():
    """sum(w) and sum((2 i + 1) w) over a row's 64-bit words mod 2^64, independent of dtype view and device."""
    t = torch.complex(torch.randn(3, 1000), torch.randn(3, 1000))
    s = sxdist.block_checksums(t).numpy().view(np.uint64)
    w = t.numpy().view(np.uint64)
    i = np.arange(1000, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    with np.errstate(over="ignore"):
        want = np.stack([w.sum(axis=1, dtype=np.uint64), (w * i).sum(axis=1, dtype=np.uint64)], axis=1)
    assert np.array_equal(s, want)
    assert torch.equal(sxdist.block_checksums(torch.view_as_real(t)), sxdist.block_checksums(t))
    with pytest.raises(ValueError):
        sxdist.block_checksums(torch.zeros(2, 3, dtype=torch.int32))


def test_single_process_gather_is_identity():
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    assert sxdist.gather_channels(t, 3) is t


def _pipe_worker(rank, world, port, n_channels, n, steps, chunks, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        sxdist.init_process_group(backend="gloo")
        lo, hi = sxdist.shard_channels(n_channels, world, rank)

        def block(step, c0, c1):          # what the rank's decimator would hold after `step`: distinct per (step, channel, sample)
            ch = torch.arange(c0, c1, dtype=torch.float32).reshape(-1, 1)
            k = torch.arange(n, dtype=torch.float32).reshape(1, -1)
            return torch.complex(1000.0 * step + ch + k / 4096.0, -(7.0 * step + ch) - k / 8192.0)

        depth = 2
        pipe = sxdist.GatherPipeline(n_channels, (hi - lo, n), torch.complex64, torch.device("cpu"), dst=0, chunks=chunks,
                                     depth=depth)
        y = [torch.empty((hi - lo, n), dtype=torch.complex64) for _ in range(depth)]
        bad = []
        for s in range(steps):
            k = s % depth
            pipe.reuse(k)
            if rank == 0 and s >= depth and not torch.equal(pipe.slot(k), block(s - depth, 0, n_channels)):
                bad.append(s - depth)
            y[k].copy_(block(s, lo, hi))          # "the kernel" overwrites the buffer the finished gather read
            pipe.submit(k, y[k])
        pipe.drain()
        if rank == 0:
            for s in range(max(0, steps - depth), steps):
                if not torch.equal(pipe.slot(s % depth), block(s, 0, n_channels)):
                    bad.append(s)
        q.put(("ok", bad, pipe.chunks, pipe.submitted))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc(), 0, 0))


@pytest.mark.parametrize("n_channels,chunks", [(16, 4), (6, 2)])
def test_pipelined_gather_two_ranks_gloo(n_channels, chunks):
    """GatherPipeline: each step's block is gathered in chunks behind the step that produced it while the next
    steps overwrite the other buffer; the root sees every step's block, whole and in global channel order."""
    world, n, steps = 2, 513, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, n_channels, n, steps, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] == "ok" for r in results), results
    for _, bad, used_chunks, submitted in results:
        assert bad == [] and submitted == steps
        assert (n_channels // world) % used_chunks == 0


def test_oracle_build_is_serialised(tmp_path):
    """Eight ranks of bench.py call oracle_lib.build() at once after a fresh checkout: the check-and-make must be
    one critical section (flock), or they race `make` over the .so files the others dlopen."""
    import shutil
    import subprocess
    code = ("import sys, os; sys.path.insert(0, %r); import oracle_lib; oracle_lib.build(); "
            "o = oracle_lib.Oracle(); print(o.ticks_to_time_ns(256, 75000.0))" % os.path.join(ROOT, "tests"))
    # a copy of oracle/ without its libraries = the fresh checkout; the tracked tree is left alone
    odir = tmp_path / "oracle"
    odir.mkdir()
    for f in ("Makefile", "sx_oracle.c", "sx_oracle.h"):
        shutil.copy(os.path.join(ROOT, "oracle", f), odir / f)
    env = dict(os.environ, SXO_ORACLE_DIR=str(odir))
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for _ in range(8)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    assert all(o[0].strip() == "3413333" for o in outs), [o[0] for o in outs]
    assert (odir / "libsxoracle.so").exists() and (odir / "libsxoracle_fast.so").exists()
