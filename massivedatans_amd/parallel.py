"""Sharding the data-set axis over the GPUs of one node (one process per GPU).

The hot path partitions trivially: spectra never interact, so rank r keeps a contiguous block
of them resident on its GPU and scores candidates against that block only.  Two small objects
are exchanged per constrained draw, both with ONE collective (KB-sized, latency-bound on xGMI):

* the shared live-point pool -- each rank contributes the unique live points of its own data
  sets; every rank needs the union to build the same RadFriends region (``allgather_pool``);
* optionally the likelihood block ``L[B, M_r]`` when the accept test runs on every rank
  (``allgather_columns``).

``torch.distributed`` is plumbing here: backend "nccl" (= RCCL over xGMI) with device tensors
on the GPU box, backend "gloo" with host tensors in the CPU tests.  The spectra themselves
are never communicated.
"""
import numpy as np


def shard_bounds(ndata, world_size):
    """Start offsets of the contiguous blocks: block r is [b[r], b[r+1]).  The first
    ``ndata % world_size`` ranks hold one spectrum more."""
    base, extra = divmod(int(ndata), int(world_size))
    sizes = [base + (1 if r < extra else 0) for r in range(world_size)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(int)


def shard_range(ndata, rank, world_size):
    b = shard_bounds(ndata, world_size)
    return int(b[rank]), int(b[rank + 1])


def local_mask(data_mask, rank, world_size):
    """The part of a global data-set mask that concerns this rank."""
    lo, hi = shard_range(len(data_mask), rank, world_size)
    return np.ascontiguousarray(data_mask[lo:hi])


def _dist():
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    return torch, dist


def _device():
    torch, dist = _dist()
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allgather_pool(local_ids, local_points):
    """Union of the ranks' live-point sets.  ``local_ids`` int64[k_r] are the global point ids
    this rank's data sets hold, ``local_points`` f64[k_r, ndim] their coordinates.  Returns
    ``(ids, points)`` with ids ascending and unique -- exactly ``numpy.unique`` over all data
    sets, the order the single-process sampler uses (multi_nested_sampler.py:134-136).

    One padded all-gather of (id, coordinates) rows; the row counts travel in the same
    message as a header row, so there is a single collective per draw."""
    torch, dist = _dist()
    world = dist.get_world_size()
    local_ids = np.asarray(local_ids, dtype=np.int64)
    local_points = np.ascontiguousarray(local_points, dtype=np.float64).reshape(len(local_ids), -1)
    ndim = local_points.shape[1]
    dev = _device()
    # capacity agreed through a max-reduce only when it has to grow (cached per process)
    cap = _capacity(len(local_ids))
    rows = np.zeros((cap + 1, ndim + 1))
    rows[0, 0] = len(local_ids)                           # header: how many rows are real
    rows[1:len(local_ids) + 1, 0] = local_ids             # ids are < 2^53: exact in f64
    rows[1:len(local_ids) + 1, 1:] = local_points
    mine = torch.from_numpy(rows).to(dev)
    # output = the ranks' blocks concatenated along dim 0 (the layout both gloo and RCCL accept)
    gathered = torch.empty((world * mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, mine)
    g = gathered.cpu().numpy().reshape(world, cap + 1, ndim + 1)
    parts = [g[r, 1:int(g[r, 0, 0]) + 1] for r in range(world)]
    allrows = np.concatenate(parts, axis=0)
    ids = allrows[:, 0].astype(np.int64)
    uniq, first = np.unique(ids, return_index=True)
    return uniq, np.ascontiguousarray(allrows[first, 1:])


_cap = 0


def _capacity(n):
    """Common row capacity of the padded gather: grows geometrically, agreed with one
    all-reduce(max) only when some rank outgrows it."""
    global _cap
    torch, dist = _dist()
    need = torch.tensor([1 if n > _cap else 0, n], dtype=torch.int64, device=_device())
    dist.all_reduce(need, op=dist.ReduceOp.MAX)
    if int(need[0]) == 1:
        _cap = max(64, 2 * int(need[1]))
    return _cap


def allgather_columns(local_block, counts=None):
    """Concatenate per-rank likelihood blocks ``L[B, M_r]`` along the data-set axis.
    ``counts`` (M_r of every rank) lets the ranks skip exchanging sizes."""
    torch, dist = _dist()
    world = dist.get_world_size()
    local_block = np.ascontiguousarray(local_block, dtype=np.float64)
    B = local_block.shape[0]
    dev = _device()
    if counts is None:
        c = torch.zeros(world, dtype=torch.int64, device=dev)
        c[dist.get_rank()] = local_block.shape[1]
        dist.all_reduce(c)
        counts = c.cpu().numpy()
    width = int(max(counts)) if len(counts) else 0
    padded = np.zeros((B, width))
    padded[:, :local_block.shape[1]] = local_block
    mine = torch.from_numpy(padded).to(dev)
    gathered = torch.empty((world * B, width), dtype=mine.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, mine)
    g = gathered.cpu().numpy().reshape(world, B, width)
    return np.concatenate([g[r, :, :int(counts[r])] for r in range(world)], axis=1)


class ShardedGaussLine(object):
    """``loglike_batch`` over all data sets, with this rank scoring only its block.
    ``backend_factory(x, y_block)`` builds the per-rank scorer (GaussLineSpectra on the GPU box;
    the tests pass an oracle-backed one)."""

    def __init__(self, x, y, backend_factory):
        torch, dist = _dist()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.ndata = y.shape[1]
        self.lo, self.hi = shard_range(self.ndata, self.rank, self.world)
        self.local = backend_factory(x, np.ascontiguousarray(y[:, self.lo:self.hi]))

    def loglike_batch(self, params, data_mask=None):
        if data_mask is None:
            data_mask = np.ones(self.ndata, dtype=bool)
        data_mask = np.asarray(data_mask, dtype=bool)
        mine = data_mask[self.lo:self.hi]
        params = np.atleast_2d(params)
        block = self.local.loglike_batch(params, mine) if mine.any() else np.zeros((len(params), 0))
        b = shard_bounds(self.ndata, self.world)
        counts = [int(data_mask[b[r]:b[r + 1]].sum()) for r in range(self.world)]
        return allgather_columns(block, counts)
