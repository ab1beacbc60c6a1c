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


class ShardedMuse(object):
    """The MUSE-style scorer (``loglike_batch(ypred[B, nx], mask)``, ``loglike_batch_lines(params[B, 5],
    mask)``: like.MuseSpectra) over all data sets, this rank scoring only its block of spectra and
    variances.  ``backend_factory(x, y_block, v_block)`` builds the per-rank scorer."""

    def __init__(self, x, y, v, backend_factory):
        torch, dist = _dist()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.ndata = y.shape[1]
        self.lo, self.hi = shard_range(self.ndata, self.rank, self.world)
        self.local = backend_factory(x, np.ascontiguousarray(y[:, self.lo:self.hi]), np.ascontiguousarray(v[:, self.lo:self.hi]))

    def _sharded(self, call, rows_in, data_mask):
        if data_mask is None:
            data_mask = np.ones(self.ndata, dtype=bool)
        data_mask = np.asarray(data_mask, dtype=bool)
        mine = data_mask[self.lo:self.hi]
        block = call(rows_in, mine) if mine.any() else np.zeros((len(rows_in), 0))
        b = shard_bounds(self.ndata, self.world)
        counts = [int(data_mask[b[r]:b[r + 1]].sum()) for r in range(self.world)]
        return allgather_columns(block, counts)

    def loglike_batch(self, ypred, data_mask=None):
        return self._sharded(self.local.loglike_batch, np.atleast_2d(ypred), data_mask)

    def loglike_batch_lines(self, params, data_mask=None):
        return self._sharded(self.local.loglike_batch_lines, np.atleast_2d(params), data_mask)


class _DeviceMemory(object):
    """Device memory owned by libmdns_hip, described through the CUDA array interface so that
    torch can wrap it without a copy (torch.as_tensor)."""

    def __init__(self, address, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(address), False),
                                         "version": 2, "strides": None}


def device_view(address, shape, typestr):
    """A torch tensor over library-owned device memory (no copy): what a collective reduces or
    gathers in place.  The library keeps ownership; the view must not outlive the allocation."""
    torch, _ = _dist()
    return torch.as_tensor(_DeviceMemory(address, shape, typestr), device=_device())


class ShardedJointState(object):
    """The joint sampler state (``jointstate``) with the data sets sharded over the ranks: rank r
    keeps live likelihoods, shelves and thresholds of ITS block next to its block of spectra.
    Per draw chunk the ranks exchange

    * one MAX all-reduce of the accept flags (B integers): every rank learns the first candidate
      that ANY data set accepts -- not the likelihoods ``L[B, M]``;
    * one all-gather of the accepted candidate's block: likelihoods + fill bits of the selected
      data sets of each rank (M numbers in all), for the host bookkeeping every rank repeats.

    Per iteration: the per-data-set minima / slots / purge decisions of ``prepare`` (two tensor
    all-gathers of fixed shape: lengths, then the padded vectors).  With the ``nccl`` backend the flags are reduced on the device
    (RCCL reduces the state's own flag buffer in place, on the library stream, which is torch's
    current stream); with ``gloo`` through the host.  Same interface as the single-process states."""

    def __init__(self, local, ndata, lo, hi):
        torch, dist = _dist()
        self.local, self.ndata, self.lo, self.hi = local, int(ndata), int(lo), int(hi)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.bounds = shard_bounds(self.ndata, self.world)
        self.nlive = local.nlive
        self.running = np.arange(self.ndata)
        self._device_flags = dist.get_backend() == "nccl" and hasattr(local, "flags_address")
        if self._device_flags:
            from . import _lib
            # the state's own flag buffer, reduced in place
            self._flags_t = device_view(local.flags_address(), (_lib.JOINT_MAX_BATCH,), "<i4")
        self.nevals_scored = 0
        self.ncalls = 0

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()

    def _mine(self, rows):
        """(local indices of the selected data sets of this rank, their number per rank)."""
        if rows is None:
            return None, np.diff(self.bounds)
        rows = np.asarray(rows)
        counts = np.diff(np.searchsorted(rows, self.bounds))
        a = int(np.searchsorted(rows, self.lo))
        return rows[a:a + counts[self.rank]] - self.lo, counts

    def init(self, xs, jitter=None):
        """``jitter`` [nlive, ndata] (the MUSE-style likelihood's noise of the initial points,
        musefuse.py:535): every rank drew the whole matrix from the common stream and keeps the
        columns of its block."""
        if jitter is not None:
            self.local.init(xs, jitter=np.ascontiguousarray(np.asarray(jitter)[:, self.lo:self.hi]))
        else:
            self.local.init(xs)
        self.ncalls += 1

    def set_running(self, running):
        self.running = np.asarray(running, dtype=int)
        mine, _ = self._mine(self.running)
        self.local.set_running(mine)

    def chunk_size(self, offered, M, hint=None):
        return self.local.chunk_size(offered, M, hint)       # from global numbers: the same on every rank

    def _gather_vectors(self, vec):
        """All-gather of one float64 vector per rank (lengths may differ): two tensor collectives
        of fixed shape -- the lengths, then the vectors padded to the longest -- instead of a
        pickled object gather.  Returns the list of the ranks' vectors."""
        torch, dist = _dist()
        vec = np.ascontiguousarray(vec, dtype=np.float64).ravel()
        n_t = torch.tensor([len(vec)], dtype=torch.int64, device=_device())
        sizes_t = torch.empty(self.world, dtype=torch.int64, device=_device())
        dist.all_gather_into_tensor(sizes_t, n_t)
        sizes = sizes_t.cpu().numpy()
        width = max(int(sizes.max()), 1)
        padded = np.zeros(width)
        padded[:len(vec)] = vec
        out_t = torch.empty(self.world * width, dtype=torch.float64, device=_device())
        dist.all_gather_into_tensor(out_t, torch.from_numpy(padded).to(_device()))
        out = out_t.cpu().numpy().reshape(self.world, width)
        return [out[r, :sizes[r]] for r in range(self.world)]

    def prepare(self):
        Lmin, arg, keep = self.local.prepare()
        n = len(Lmin)
        # one vector per rank: [kept-matrix width (0: nothing was dropped here), Lmin, argmin,
        # kept bits row by row]; small integers are exact in float64
        width = 0 if keep is None else keep.shape[1]
        mine = np.concatenate(([float(width)], Lmin, np.asarray(arg, dtype=np.float64),
                               np.zeros(0) if keep is None else keep.astype(np.float64).ravel()))
        parts = self._gather_vectors(mine)
        counts = [(len(p) - 1) // (2 + int(p[0])) for p in parts]
        Lmin = np.concatenate([p[1:1 + c] for p, c in zip(parts, counts)])
        arg = np.concatenate([p[1 + c:1 + 2 * c] for p, c in zip(parts, counts)]).astype(int)
        widths = [int(p[0]) for p in parts]
        if max(widths) == 0:
            return Lmin, arg, None
        # a rank that dropped nothing reports no matrix: all of its entries stay (the sampler masks
        # the columns past each shelf's length)
        keep = np.ones((len(Lmin), max(widths)), dtype=bool)
        at = 0
        for p, c, w in zip(parts, counts, widths):
            if w:
                keep[at:at + c, :] = False
                keep[at:at + c, :w] = p[1 + 2 * c:].reshape(c, w) != 0
            at += c
        return Lmin, arg, keep

    def _reduce_flags(self, B):
        torch, dist = _dist()
        if self._device_flags:
            dist.all_reduce(self._flags_t[:B], op=dist.ReduceOp.MAX)
            return self._flags_t[:B].cpu().numpy()
        flags = torch.from_numpy(np.ascontiguousarray(self._local_flags, dtype=np.int32)).to(_device())
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        flags = flags.cpu().numpy()
        if hasattr(self.local, "set_flags"):
            self.local.set_flags(flags)
        return flags

    @property
    def nparams(self):
        return getattr(self.local, "nparams", 3)

    def took(self, rows, beats):
        """(a native constrainer's draw ended in this state's ``draw_params``: nothing to add)"""

    def draw(self, xs, rows):
        return self.draw_params(self.local.to_kernel_params(xs), rows)

    def draw_params(self, params, rows, jitter=None):
        """``draw`` for candidates given as kernel parameter rows: what a native constrainer hands over
        (``constrainer.python_backend``); every rank runs the same constrainer on the same random
        stream and meets the others here."""
        torch, dist = _dist()
        mine, counts = self._mine(rows)
        xs = params
        B = len(xs)
        if hasattr(self.local, "score_backend"):
            return self._draw_halves(params, mine, counts, jitter)
        if jitter is not None:
            # every rank's constrainer drew the noise of ALL selected data sets from the common stream
            # (candidate by candidate, musefuse.py:535): this rank adds the columns of its own
            at = int(counts[:self.rank].sum())
            mine_jitter = np.ascontiguousarray(np.asarray(jitter)[:, at:at + int(counts[self.rank])])
            self._local_flags = self.local.score_params(params, mine, jitter=mine_jitter)
        else:
            self._local_flags = self.local.score_params(params, mine)
        if self._local_flags is None and not self._device_flags:
            self._local_flags = self.local.flags()
        self.ncalls += 1
        self.nevals_scored += B * int(counts.sum())
        flags = self._reduce_flags(B)
        hit = np.flatnonzero(flags)
        if len(hit) == 0:
            return -1, None, None, B
        idx = int(hit[0])
        Lrow, beats = self.local.commit(idx)
        # the accepted candidate's block of every rank: likelihoods (when the local state hands
        # them over), then fill bits
        width = int(counts.max())
        block = np.zeros(2 * width)
        if Lrow is not None:
            block[:len(Lrow)] = Lrow
        block[width:width + len(beats)] = beats
        mine_t = torch.from_numpy(block).to(_device())
        gathered = torch.empty(self.world * 2 * width, dtype=mine_t.dtype, device=_device())
        dist.all_gather_into_tensor(gathered, mine_t)
        g = gathered.cpu().numpy().reshape(self.world, 2, width)
        L = np.concatenate([g[r, 0, :counts[r]] for r in range(self.world)]) if Lrow is not None else None
        b = np.concatenate([g[r, 1, :counts[r]] for r in range(self.world)]) != 0
        return idx, L, b, B

    # ---- the chunk in two halves on the device (local states on the GPU) -------------------------
    def _direct(self):
        """RCCL called directly on the library's stream (massivedatans_amd.rccl), or None: backend not
        nccl, library not bindable, or MDNS_SHARDED_COLLECTIVES=torch."""
        if not hasattr(self, "_rccl"):
            import os
            self._rccl = None
            torch, dist = _dist()
            if dist.get_backend() == "nccl" and os.environ.get("MDNS_SHARDED_COLLECTIVES", "rccl") != "torch":
                try:
                    from . import _lib, rccl
                    comm = rccl.from_torch_distributed()
                    stream = _lib.require_device().mdns_get_stream()
                    self._rccl = (rccl, comm, stream)
                    self._allbits = None
                except Exception:       # noqa: BLE001 -- the torch path is always there
                    self._rccl = None
        return self._rccl

    def _draw_halves(self, params, mine, counts, jitter):
        """One chunk: this rank scores its selected data sets (votes on the device), the ranks MAX-reduce
        the votes -- in place, on the stream the kernels run on -- every rank commits the first candidate
        that has a vote, and the fill bits of all ranks are gathered for the bookkeeping every rank keeps."""
        torch, dist = _dist()
        from . import _lib
        B = len(params)
        mine_jitter = None
        if jitter is not None:
            at = int(counts[:self.rank].sum())
            mine_jitter = np.ascontiguousarray(np.asarray(jitter)[:, at:at + int(counts[self.rank])])
        self.local.score_backend(params, mine, mine_jitter)
        self.ncalls += 1
        self.nevals_scored += B * int(counts.sum())
        direct = self._direct()
        if direct is not None:
            rccl, comm, stream = direct
            comm.all_reduce(self.local.votes_address(), self.local.votes_address(), B, rccl.INT32, rccl.MAX, stream)
        elif dist.get_backend() == "nccl":
            votes_t = device_view(self.local.votes_address(), (_lib.JOINT_MAX_BATCH,), "<i4")
            dist.all_reduce(votes_t[:B], op=dist.ReduceOp.MAX)
        else:
            votes = torch.from_numpy(self.local.votes())
            dist.all_reduce(votes, op=dist.ReduceOp.MAX)
            self.local.set_votes(votes.numpy())
        idx, beats = self.local.commit_backend()
        # (every rank found the same first voted candidate; -1: nobody accepted anything)
        if idx < 0:
            return -1, None, None, B
        width = int(counts.max())
        block = np.zeros(width)
        block[:len(beats)] = beats
        mine_t = torch.from_numpy(block).to(_device())
        gathered = torch.empty(self.world * width, dtype=mine_t.dtype, device=_device())
        dist.all_gather_into_tensor(gathered, mine_t)
        g = gathered.cpu().numpy().reshape(self.world, width)
        b = np.concatenate([g[r, :counts[r]] for r in range(self.world)]) != 0
        return idx, None, b, B

    def advance(self):
        self.local.advance()

    def live_matrix(self):
        parts = self._gather_vectors(np.ascontiguousarray(self.local.live_matrix()).ravel())
        # (a rank that owns no running data set contributes an empty block: its width is stated, not inferred)
        return np.concatenate([p.reshape(self.nlive, len(p) // self.nlive) for p in parts], axis=1)

    def thresholds(self):
        higher, n = self.local.thresholds()
        parts = self._gather_vectors(np.concatenate((higher, np.asarray(n, dtype=np.float64))))
        return (np.concatenate([p[:len(p) // 2] for p in parts]),
                np.concatenate([p[len(p) // 2:] for p in parts]).astype(int))


class LocalColumns(object):
    """What the evidence integration sees of a sampler whose data sets are spread over the ranks
    (``ShardedJointState``): THIS rank's data sets only.  The sampler itself stays whole on every rank
    -- its draws are collective -- but everything the integrator does is per data set
    (multi_nested_integrator.py:26-175: one evidence, one information, one stopping test per column), so
    each rank integrates its block: 1/N of the per-iteration host work, and the live likelihoods are read
    from the rank's own joint state instead of being gathered every 50 iterations (80 MB at 100 000 data
    sets x 100 live points).  Two things are collective: which data sets are finished (``cut_down``: one
    all-gather of flags per check) and whether everybody is (``everybody_done``: one all-reduce).
    ``gather`` puts the per-data-set results of all ranks together at the end."""

    collective_checks = True

    def __init__(self, sampler):
        torch, dist = _dist()
        self.sampler = sampler
        self.joint = sampler.joint
        if not isinstance(self.joint, ShardedJointState):
            raise TypeError("LocalColumns needs a sampler over a ShardedJointState")
        self.nlive_points = sampler.nlive_points
        self.lo, self.hi = self.joint.lo, self.joint.hi
        self._running = np.arange(self.joint.ndata)               # global ids of the data sets still sampled
        self._live = None
        self._select()

    # ---- what touches the other ranks / this rank's state (tests replace these three) ----
    def _min_over_ranks(self, values):
        """Element-wise minimum of an int32 vector over the ranks."""
        torch, dist = _dist()
        t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.int32)).to(_device())
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.cpu().numpy()

    def _local_live(self):
        """[nlive, this rank's running data sets], fetched from the joint state once per iteration"""
        if self._live is None:
            self._live = np.asarray(self.joint.local.live_matrix())
        return self._live

    def _gather_columns(self, row):
        return allgather_columns(np.asarray(row, dtype=np.float64)[None, :])[0]

    def _select(self):
        self._mine = np.flatnonzero((self._running >= self.lo) & (self._running < self.hi))   # places in the sampler's running list
        #: (numpy sums a lone column pairwise, the columns of a wider matrix row by row: see _column_sums)
        self.sums_row_by_row = len(self._mine) == 1 and len(self._running) > 1

    @property
    def ndata(self):
        return len(self._mine)

    def __getattr__(self, name):                                   # ndraws, nevals, native, ... of the whole sampler
        return getattr(self.sampler, name)

    def __next__(self):
        u, x, L = next(self.sampler)
        self._live = None
        return u[self._mine], x[self._mine], np.asarray(L)[self._mine]

    next = __next__

    @property
    def Lmax(self):
        return self._local_live().max(axis=0)

    def remainder_likelihoods(self):
        return np.ascontiguousarray(np.sort(self._local_live(), axis=0))

    def remainder_arrays(self, d):
        """(u[nlive, ndim], x[nlive, ndim], L[nlive]) of this rank's d-th running data set, ascending."""
        col = self._local_live()[:, d]
        order = np.argsort(col)
        p = self.sampler.live_pointsp[order, self._mine[d]]
        return self.sampler.pointpile[p], self.sampler.pointpilex[p], col[order]

    def remainder_arrays_many(self, ds):
        ds = np.asarray(ds, dtype=int)
        L = self._local_live()[:, ds]
        order = np.argsort(L, axis=0)
        cols = np.arange(len(ds))[None, :]
        p = np.asarray(self.sampler.live_pointsp)[:, self._mine[ds]][order, cols]
        return self.sampler.pointpile[p], self.sampler.pointpilex[p], L[order, cols]

    def cut_down(self, surviving):
        """``surviving``: flags of this rank's running data sets; every rank calls it at every check."""
        flags = np.ones(len(self._running), dtype=np.int32)
        flags[self._mine] = np.asarray(surviving, dtype=np.int32)
        # (a data set belongs to one rank: the others left their 1 there)
        keep = self._min_over_ranks(flags).astype(bool)
        if not keep.all():
            self.sampler.cut_down(keep)
            self._running = self._running[keep]
            self._live = None
            self._select()

    def everybody_done(self, mine_done):
        return bool(self._min_over_ranks(np.array([1 if mine_done else 0]))[0])

    def gather(self, results):
        """The ranks' per-data-set results side by side, in the order of the data sets (the blocks are
        contiguous): ``logZ``, ``logZerr``, ``information`` of ALL data sets on every rank.  ``weights`` -- the
        posterior samples, [iterations][u, x, L, w, mask] over this rank's columns -- stay where they are
        (``columns`` says which)."""
        out = dict(results)
        for key in ("logZ", "logZerr", "information"):
            out[key] = self._gather_columns(results[key])
        out["columns"] = (self.lo, self.hi)
        return out
