"""Multi-GPU layout of the resampling path: independent channels are sharded
one contiguous group per GPU (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).  The FIR phase needs no exchange at all;
the only collective is the gather of the decimated output to a root rank.

The reference is single-device and single-channel (getNumChannels returns 1,
SoapySX.cpp:1591-1595); this module is new.
"""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_channels(n_channels, world_size, rank):
    """Contiguous, balanced channel range [lo, hi) owned by `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(n_channels, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a
    single process).  Returns (rank, local_rank, world_size)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)        # one GPU per rank
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def gather_channels(local, n_channels, dst=0, group=None, always_collective=False):
    """Gather every rank's [local_channels, n] block into a [n_channels, n]
    tensor on `dst` (channel-major, in global channel order); other ranks get
    None.  Shards may differ in size by one channel; blocks are padded to the
    largest shard for the collective and trimmed on the root.  A world of one
    rank returns `local` itself unless always_collective is set (which runs the
    collective anyway: a 1-GPU box can then exercise the RCCL call)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_channels(n_channels, world, rank)
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d channels, expected %d" % (rank, local.shape[0], hi - lo))
    widest = (n_channels + world - 1) // world
    send = local
    if local.shape[0] < widest:
        send = torch.zeros((widest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    # complex tensors travel as their real view (RCCL has no complex dtype)
    is_complex = send.is_complex()
    wire = torch.view_as_real(send) if is_complex else send
    wire = wire.contiguous()
    full = None
    bufs = None
    if rank == dst:
        # one destination tensor, the collective writes straight into its per-rank slices
        full = torch.empty((world * widest,) + tuple(wire.shape[1:]), dtype=wire.dtype, device=wire.device)
        bufs = list(full.split(widest, dim=0))
    dist.gather(wire, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if n_channels == world * widest:                      # equal shards: already in global channel order
        out = full
    else:
        parts = []
        for r in range(world):
            rlo, rhi = shard_channels(n_channels, world, r)
            parts.append(bufs[r][: rhi - rlo])
        out = torch.cat(parts, dim=0)
    return torch.view_as_complex(out) if is_complex else out
