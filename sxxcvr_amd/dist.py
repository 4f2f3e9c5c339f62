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


# ------------------------------------------------------------------------------------------------------------
# Certifying a gather: every rank states what it sent, the root recomputes it over what arrived
# ------------------------------------------------------------------------------------------------------------
def block_checksums(block):
    """Per channel of a [channels, ...] block (complex64 samples, or any dtype whose row is a whole number of 64-bit
    words): two 64-bit sums over the row's words w[i] taken as integers, mod 2^64 -- sum(w[i]) and the position-weighted
    sum((2 i + 1) w[i]).  One flipped bit changes the first, two exchanged samples the second, a block that is another
    channel's (or another step's) both.  Returns int64 [channels, 2] on the block's device; computed with the tensor
    library's wrapping integer arithmetic, so a GPU and a host evaluation of the same bytes agree."""
    import torch
    w = torch.view_as_real(block) if block.is_complex() else block
    w = w.contiguous().reshape(block.shape[0], -1)
    if (w.shape[1] * w.element_size()) % 8:
        raise ValueError("a channel's row must be a whole number of 64-bit words")
    w = w.view(torch.int64)
    weight = torch.arange(w.shape[1], device=w.device, dtype=torch.int64) * 2 + 1
    out = torch.empty((w.shape[0], 2), dtype=torch.int64, device=w.device)
    for c in range(w.shape[0]):                       # per channel: bounds the temporaries to one row
        out[c, 0] = w[c].sum()
        out[c, 1] = (w[c] * weight).sum()
    return out


def exchange_checksums(local_sums, n_channels, group=None):
    """All-gather the ranks' [local_channels, 2] checksum tables into the [n_channels, 2] table of the whole job, in
    global channel order (equal shards); returns a host tensor.  16 bytes per channel.  bench.py hands its host-side
    control group (gloo), so that the data-path communicator carries nothing but the data; over an nccl group (the
    fallback when no gloo group could be made) the tables travel as device tensors."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if n_channels % world:
        raise ValueError("equal shards only (%d channels over %d ranks)" % (n_channels, world))
    local = local_sums.detach().to(torch.int64).contiguous()
    if tuple(local.shape) != (n_channels // world, 2):
        raise ValueError("a rank states %s checksums, expected %s" % (tuple(local.shape), (n_channels // world, 2)))
    if dist.get_backend(group) == "nccl":
        local = local.to(torch.device("cuda", torch.cuda.current_device()))
    else:
        local = local.cpu()
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.cat(parts, dim=0).cpu()


def check_gathered(full, stated, skip=()):
    """Root side: recompute the checksums over the gathered [n_channels, ...] tensor and compare with what the
    senders stated.  Returns the list of channels that differ (empty = every block arrived whole and in its place);
    `skip`: channels not to hold against the gather (the root's own, which never travelled, are still checked by
    default)."""
    got = block_checksums(full).cpu()
    bad = (got != stated.cpu()).any(dim=1).nonzero().flatten().tolist()
    return [c for c in bad if c not in set(skip)]


class GatherPipeline:
    """The exchange step in steady state ("chunk and overlap with the next tile's compute"):
    every step's decimated output is gathered to `dst` in `chunks` sub-collectives (groups of whole channels, so
    each send buffer is contiguous) that run BEHIND the kernel that produced the block and BESIDE the kernels of the
    next steps.  The caller alternates between `depth` output buffers:

        for s in range(steps):
            pipe.reuse(s % depth)            # the gather that last read buffer s % depth is done (GPU-side wait)
            plan.process(x, out=y[s % depth])
            pipe.submit(s % depth, y[s % depth])
        pipe.drain()

    backend nccl (= RCCL over xGMI): dist.gather(async_op=True) per chunk; the collective is ordered behind the
    current stream's work by ProcessGroupNCCL and runs on its own stream; reuse() makes the current stream wait for
    it, the host never blocks.  Any other backend (gloo: host tensors; the dry-run stand-in for a 1-GPU box and the
    CPU tests): the block is staged to the host behind the kernel and its collectives are issued one step late, when
    the staging copy has certainly been queued; same results, same order.

    On `dst`, slot(k) is the [n_channels, ...] tensor that buffer k's last gather filled (global channel order);
    the reference is single-channel (SoapySX.cpp:1591-1595), this layout is the build's."""

    def __init__(self, n_channels, local_shape, dtype, device, dst=0, chunks=2, depth=2, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_channels, self.depth = n_channels, depth
        if n_channels % self.world:
            raise ValueError("the pipelined gather takes equal shards (%d channels over %d ranks)" % (n_channels, self.world))
        self.local_ch = n_channels // self.world
        if tuple(local_shape)[0] != self.local_ch:
            raise ValueError("a rank holds %d channels, expected %d" % (local_shape[0], self.local_ch))
        chunks = max(1, min(int(chunks), self.local_ch))
        while self.local_ch % chunks:
            chunks -= 1
        self.chunks, self.cpc = chunks, self.local_ch // chunks         # channels per chunk
        self.host = dist.get_backend(group) != "nccl"
        self.is_complex = dtype.is_complex
        wire_dtype = {torch.complex64: torch.float32, torch.complex128: torch.float64}.get(dtype, dtype)
        tail = tuple(local_shape[1:]) + ((2,) if self.is_complex else ())
        cdev = torch.device("cpu") if self.host else device
        self.full = None
        if self.rank == dst:
            # [depth][world, local_ch, ...]: rank r's chunk j lands in full[k][r, j*cpc:(j+1)*cpc]
            self.full = [torch.empty((self.world, self.local_ch) + tail, dtype=wire_dtype, device=cdev) for _ in range(depth)]
        self.stage = None
        if self.host:
            self.stage = [torch.empty((self.local_ch,) + tail, dtype=wire_dtype).pin_memory() if torch.cuda.is_available()
                          else torch.empty((self.local_ch,) + tail, dtype=wire_dtype) for _ in range(depth)]
        self.works = [[] for _ in range(depth)]
        self.pending = []                                               # host mode: (slot, event) not yet issued
        self.submitted = 0

    def _wire(self, t):
        return self.torch.view_as_real(t) if self.is_complex else t

    def _issue(self, k, wire):
        for j in range(self.chunks):
            send = wire[j * self.cpc:(j + 1) * self.cpc]
            bufs = None
            if self.rank == self.dst:
                bufs = [self.full[k][r, j * self.cpc:(j + 1) * self.cpc] for r in range(self.world)]
            self.works[k].append(self.dist.gather(send, bufs, dst=self.dst, group=self.group, async_op=True))

    def _pump(self, force=False):
        while self.pending:
            k, ev = self.pending[0]
            if ev is not None:
                if not force and len(self.pending) < 2 and not ev.query():
                    return
                ev.synchronize()
            self.pending.pop(0)
            self._issue(k, self.stage[k])

    def reuse(self, k):
        """Buffer k (and slot k on dst) may be overwritten once this returns (nccl: once the current stream gets
        there)."""
        if self.host:
            self._pump(force=any(p[0] == k for p in self.pending))
        for w in self.works[k]:
            w.wait()
        self.works[k] = []

    def submit(self, k, local):
        """Queue the gather of `local` (this step's output, produced on the current stream) into slot k."""
        wire = self._wire(local)
        if not wire.is_contiguous():
            raise ValueError("the step's output must be contiguous [channels, samples]")
        if self.host:
            ev = None
            if wire.is_cuda:
                self.stage[k].copy_(wire, non_blocking=True)
                ev = self.torch.cuda.Event()
                ev.record()
            else:
                self.stage[k].copy_(wire)
            self.pending.append((k, ev))
            self._pump()
        else:
            self._issue(k, wire)
        self.submitted += 1

    def drain(self):
        if self.host:
            self._pump(force=True)
        for k in range(self.depth):
            for w in self.works[k]:
                w.wait()
            self.works[k] = []
        if not self.host and self.torch.cuda.is_available():
            self.torch.cuda.synchronize()

    def slot(self, k):
        """dst only: the gathered [n_channels, ...] block of buffer k's last submit (after reuse(k) / drain())."""
        if self.full is None:
            return None
        t = self.full[k].reshape((self.n_channels,) + tuple(self.full[k].shape[2:]))
        return self.torch.view_as_complex(t) if self.is_complex else t
