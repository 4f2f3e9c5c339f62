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


def test_single_process_gather_is_identity():
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    assert sxdist.gather_channels(t, 3) is t
