"""The collectives of the multi-GPU layout through RCCL itself (torch.distributed backend "nccl") on the GPUs this
box has.  With one GPU the world is one rank: communicator set-up, the gather of the decimated channels in the
wire layout bench.py uses, the MAX all-reduce of the timing and the barrier all run through librccl -- what a
1-GPU box can prove about the N > 1 path beyond the gloo tests.  With two or more GPUs the same script runs one
rank per GPU over xGMI."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
import sxxcvr_amd
from sxxcvr_amd import dist as sxdist
from sxxcvr_amd.resampler import DECIMATE

rank, local_rank, world = sxdist.env_rank()
torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", rank=rank, world_size=world)
assert dist.get_backend() == "nccl"
per, n_in = 8, 1 << 16
lo, hi = sxdist.shard_channels(per * world, world, rank)
taps = sxxcvr_amd.design_lowpass(128, 4)
plan = sxxcvr_amd.Resampler(DECIMATE, taps, 4, nchan=per, device=local_rank)
x = torch.empty((per, n_in), dtype=torch.complex64, device="cuda")
sxxcvr_amd.synth_fill(x, 0x51255, first_channel=lo, start=0)
y = plan.process(x)
torch.cuda.synchronize()
full = sxdist.gather_channels(y, per * world, dst=0, always_collective=True)
t = torch.tensor([1.0 + rank], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == float(world)
if rank == 0:
    assert full.shape == (per * world, n_in // 4) and full.dtype == torch.complex64
    assert torch.equal(full[:per], y)
    # every rank's block is what this rank computes for those channels
    for r in range(1, world):
        sxxcvr_amd.synth_fill(x, 0x51255, first_channel=per * r, start=0)
        plan.reset()
        want = plan.process(x)
        torch.cuda.synchronize()
        assert torch.equal(full[per * r:per * (r + 1)], want), r
    print("rccl ok world", world)

# what bench.py's GatherCertifier does: senders state per-channel checksums computed on their GPU, the tables travel over
# a host-side gloo group created BESIDE the nccl one, the root recomputes them over the gathered tensor
ctl = dist.new_group(backend="gloo")
sums = sxdist.block_checksums(y)
assert sums.is_cuda and tuple(sums.shape) == (per, 2)
stated = sxdist.exchange_checksums(sums, per * world, group=ctl)
assert tuple(stated.shape) == (per * world, 2) and not stated.is_cuda
full = sxdist.gather_channels(y, per * world, dst=0, always_collective=True)
if rank == 0:
    assert sxdist.check_gathered(full, stated) == []
    assert torch.equal(sxdist.block_checksums(full.cpu()), sxdist.block_checksums(full).cpu())      # host and GPU evaluation agree
    hit = per * (world - 1) + 3                                     # a channel of the last rank's block
    full.view(torch.int64)[hit, 777] ^= 1
    assert sxdist.check_gathered(full, stated) == [hit]
    print("rccl checksums ok world", world)
ids = [None] * world
dist.all_gather_object(ids, torch.cuda.get_device_properties(local_rank).name, group=ctl)
assert len(ids) == world and dist.get_world_size() == world
dist.barrier(group=ctl)

# the pipelined gather (GatherPipeline) over nccl: async collectives behind the kernels, two buffers in turn;
# every step's gathered block must be that step's output (steps differ: the stream moves on)
depth, steps = 2, 5
pipe = sxdist.GatherPipeline(per * world, (per, n_in // 4), torch.complex64, torch.device("cuda", local_rank), dst=0,
                             chunks=4, depth=depth)
assert not pipe.host and pipe.chunks == 4
plan.reset()
sxxcvr_amd.synth_fill(x, 0x51255, first_channel=lo, start=0)
ys = [torch.empty((per, n_in // 4), dtype=torch.complex64, device="cuda") for _ in range(depth)]
keep, seen = [], {}
for s in range(steps):
    k = s %% depth
    pipe.reuse(k)
    if rank == 0 and s >= depth:
        seen[s - depth] = pipe.slot(k)[:per].clone()       # ordered behind the gather by reuse()'s stream wait
    plan.process(x, out=ys[k])
    keep.append(ys[k].clone())
    pipe.submit(k, ys[k])
pipe.drain()
if rank == 0:
    for s in range(steps - depth, steps):
        seen[s] = pipe.slot(s %% depth)[:per].clone()
    torch.cuda.synchronize()
    for s in range(steps):
        assert torch.equal(seen[s], keep[s]), s
    assert not torch.equal(keep[0], keep[1])
    print("rccl pipeline ok world", world)
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_allreduce_barrier_through_rccl(tmp_path):
    import torch
    world = min(max(torch.cuda.device_count(), 1), 2)
    script = tmp_path / "rccl_probe.py"
    script.write_text(SCRIPT % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "rccl ok world %d" % world in outs[0]
    assert "rccl checksums ok world %d" % world in outs[0], outs[0][-1500:]
    assert "rccl pipeline ok world %d" % world in outs[0], outs[0][-1500:]


def _run(cmd, env=None, timeout=300):
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    return p.returncode, p.stdout


def test_c_caller_gathers_through_the_c_abi(tmp_path):
    """sxfir_comm_* (include/sxfir.h) from plain C, no Python and no torch.distributed in the processes that move
    the data: tools/gather_c.c decimates 8 channels per rank and gathers the blocks to rank 0 over librccl -- rank per
    process (the id travels through a file) on min(visible GPUs, 2) ranks, and one process driving all its
    communicators (sxfir_comm_init_all / sxfir_comm_gather_all).  With one GPU the world is one rank: communicator
    set-up, the root's own block, the chunking and the checks all run; with more, over xGMI."""
    import json
    import torch
    exe = os.path.join(ROOT, "sxxcvr_amd", "lib", "sx_gather_c")
    assert os.path.exists(exe), "sx_gather_c is missing: python -m sxxcvr_amd.build"
    ngpu = max(torch.cuda.device_count(), 1)
    world = min(ngpu, 2)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    idfile = str(tmp_path / "rccl_id")
    procs = [subprocess.Popen([exe, "ranks", str(world), str(r), idfile, "18", "3"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    line = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][-1])
    assert line["verified"] is True and line["nranks"] == world and line["bytes_per_rank"] == 8 * 8 * (1 << 16)
    rc, out = _run([exe, "all", str(world), "18", "3"], env=env)
    assert rc == 0, out[-1500:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["verified"] is True and line["form"].startswith("one process")


def test_comm_gather_from_python_world_of_one():
    """The same entry points through ctypes on a world of one rank: the root's own block is copied into its place in
    the gathered buffer (or left alone when it is already there), chunked and unchunked, on the caller's stream;
    argument errors come back as SXFIR_EINVAL with a message."""
    import ctypes as C
    import numpy as np
    import torch
    import sxxcvr_amd
    lib = sxxcvr_amd.load_sxfir()
    ident = (C.c_ubyte * 128)()
    assert lib.sxfir_comm_unique_id(ident) == 0, lib.sxfir_last_error()
    comm = C.c_void_p()
    assert lib.sxfir_comm_init_rank(C.byref(comm), ident, 1, 0, -1) == 0, lib.sxfir_last_error()
    rank, n, dev = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    assert lib.sxfir_comm_rank(comm, C.byref(rank), C.byref(n), C.byref(dev)) == 0
    assert (rank.value, n.value) == (0, 1) and dev.value == torch.cuda.current_device()
    # ... and as RCCL itself reports them (ncclCommUserRank / ncclCommCount / ncclCommCuDevice): what bench.py records
    rank, n, dev = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    assert lib.sxfir_comm_query(comm, C.byref(rank), C.byref(n), C.byref(dev)) == 0, lib.sxfir_last_error()
    assert (rank.value, n.value) == (0, 1) and dev.value == torch.cuda.current_device()
    assert lib.sxfir_comm_query(None, None, None, None) == -1
    bdf = C.create_string_buffer(32)
    assert lib.sxfir_device_pci_bus_id(-1, bdf, 32) == 0, lib.sxfir_last_error()
    assert bdf.value.count(b":") == 2 and b"." in bdf.value, bdf.value
    assert lib.sxfir_device_pci_bus_id(0, bdf, 8) == -1                    # buffer too small: refused, not truncated
    src = torch.arange(1 << 16, dtype=torch.int32, device="cuda")
    dst = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    nbytes = src.numel() * 4
    for chunk in (0, 4096, 100000):
        dst.zero_()
        assert lib.sxfir_comm_gather(comm, src.data_ptr(), dst.data_ptr(), nbytes, nbytes, 0, chunk, st) == 0, lib.sxfir_last_error()
        torch.cuda.synchronize()
        assert torch.equal(src, dst), chunk
    assert lib.sxfir_comm_gather(comm, src.data_ptr(), src.data_ptr(), nbytes, nbytes, 0, 0, st) == 0     # in place
    assert lib.sxfir_comm_gather(comm, src.data_ptr(), dst.data_ptr(), nbytes, nbytes, 1, 0, st) == -1    # no such root
    assert b"root" in lib.sxfir_last_error()
    assert lib.sxfir_comm_gather(comm, src.data_ptr(), dst.data_ptr(), nbytes, nbytes - 4, 0, 0, st) == -1
    assert lib.sxfir_comm_destroy(comm) == 0
