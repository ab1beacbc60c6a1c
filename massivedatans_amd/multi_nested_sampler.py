"""Joint nested sampler over many data sets sharing one parameter space.

Host-side orchestration with the behaviour of the reference's
``multi_nested_sampler.py:49-570`` (``MultiNestedSampler``): the same constructor, iterator
protocol (``next(sampler) -> (u[ndata,ndim], x[ndata,ndim], L[ndata])``), ``remainder`` /
``cut_down`` methods, attributes (``nlive_points``, ``ndata``, ``ndraws``, ``Lmax``, ...) and --
given bit-identical likelihood values -- the same integer bookkeeping (live-point id matrix,
shelves, superpoints, data-set groups) and the same consumption of the global legacy numpy RNG.

State (names as in the reference):
  pointpile[npoints, ndim], pointpilex   every point ever accepted, unit-cube / physical
  live_pointsp[nlive, ndata] int         id (row of pointpile) of each live point of each data set
  live_pointsL[nlive, ndata] f64         its likelihood for that data set
  shelves                                per data set a FIFO of accepted (id, L) pairs waiting to
                                         replace its worst live point (array-backed, see _Shelves;
                                         u and x of an entry are pointpile[id], pointpilex[id])
  superpoints                            ids still shared by ALL data sets

What differs from the reference is mechanical: no printing (``logging`` at DEBUG), the initial
``nlive`` likelihood vectors come from one batched launch, the constrainers can hand whole
chunks of proposed candidates to a joint state (``draw_batch``) or be native objects, and the
per-data-set Python loops of the reference (shelf lists, ``numpy.unique`` over the whole id
matrix every iteration) are array operations: at 10 000 data sets they, not the likelihood,
set the wall-clock.  None of this changes a single value (tests/test_orchestration.py).
"""
import ctypes
import logging
import time
from collections import defaultdict

import numpy

log = logging.getLogger("massivedatans_amd")

from . import _host

_DEBUG_SHELVES = __import__("os").environ.get("MDNS_DEBUG_SHELVES") == "1"


def _host_lib():
    """``libmdns_host.so`` (csrc/host_groups.c, plain C): the grouping walk as native host code.
    Optional -- without it the same walk runs in Python (``_walk_python``), 10-30x slower."""
    return _host.lib()


def find_nsmallest(n, arr1, arr2):
    """(n+1)-th smallest value of the two arrays together (multi_nested_sampler.py:44-47)."""
    merged = numpy.concatenate((arr1, arr2))
    return numpy.partition(merged, n)[n]


class _Shelves(object):
    """FIFO queues of (point id, L), one per data set, in two padded arrays.  Same behaviour
    as the reference's ``self.shelves`` list of lists of ``(pj, uj, xj, Lj)`` tuples
    (multi_nested_sampler.py:117,137-138,482-485,513): append at the tail, pop at the head,
    purge keeps order."""

    def __init__(self, ndata, cap=4):
        self.n = numpy.zeros(ndata, dtype=int)
        self.p = numpy.zeros((ndata, cap), dtype=int)
        self.L = numpy.full((ndata, cap), numpy.inf)

    def __len__(self):
        return len(self.n)

    def empty(self):
        return self.n == 0

    def purge(self, Lmins):
        """Drop entries that no longer beat their data set's threshold, keeping the order."""
        cap = self.p.shape[1]
        valid = numpy.arange(cap)[None, :] < self.n[:, None]
        keep = valid & (self.L > Lmins[:, None])
        if keep.sum() == valid.sum():
            return
        self._compact(keep)

    def purge_kept(self, kept):
        """The same with the decision made elsewhere (the joint state on the device):
        ``kept[d, e]`` says whether entry e of data set d stays."""
        cap = self.p.shape[1]
        valid = numpy.arange(cap)[None, :] < self.n[:, None]
        keep = numpy.zeros((len(self.n), cap), dtype=bool)
        keep[:, :kept.shape[1]] = kept[:, :cap]
        self._compact(keep & valid)

    def _compact(self, keep):
        cap = self.p.shape[1]
        order = numpy.argsort(~keep, axis=1, kind='stable')       # kept entries first, in order
        self.p = numpy.take_along_axis(self.p, order, axis=1)
        self.L = numpy.take_along_axis(self.L, order, axis=1)
        self.n = keep.sum(axis=1)
        self.L[numpy.arange(cap)[None, :] >= self.n[:, None]] = numpy.inf

    def append(self, rows, pid, Ls):
        if len(rows) == 0:
            return
        if (self.n[rows] >= self.p.shape[1]).any():
            extra = self.p.shape[1]
            self.p = numpy.hstack((self.p, numpy.zeros((len(self.n), extra), dtype=int)))
            self.L = numpy.hstack((self.L, numpy.full((len(self.n), extra), numpy.inf)))
        pos = self.n[rows]
        self.p[rows, pos] = pid
        self.L[rows, pos] = Ls
        self.n[rows] += 1

    def pop_heads(self):
        """Head (id, L) of every queue; every queue must be non-empty."""
        assert (self.n > 0).all()
        pid, L = self.p[:, 0].copy(), self.L[:, 0].copy()
        m = int(self.n.max())                           # nothing waits beyond column m - 1
        self.p[:, :m - 1] = self.p[:, 1:m]
        self.L[:, :m - 1] = self.L[:, 1:m]
        self.L[:, m - 1] = numpy.inf
        self.n -= 1
        return pid, L

    def select(self, surviving):
        self.n, self.p, self.L = self.n[surviving], self.p[surviving], self.L[surviving]

    def as_lists(self, pointpile, pointpilex):
        """The reference's representation (for inspection)."""
        return [[(int(self.p[d, k]), pointpile[self.p[d, k]], pointpilex[self.p[d, k]], self.L[d, k])
                 for k in range(self.n[d])] for d in range(len(self.n))]


class _LazyMask(object):
    """The boolean mask over ``n`` data sets that is True at ``members`` -- built when somebody looks
    at it (the native draw path works from the member indices and never does)."""

    def __init__(self, n, members):
        self.n, self.members, self._mask = n, members, None

    def __array__(self, dtype=None, copy=None):
        if self._mask is None:
            self._mask = numpy.zeros(self.n, dtype=bool)
            self._mask[self.members] = True
        return self._mask if dtype is None else self._mask.astype(dtype)

    def __len__(self):
        return self.n

    def sum(self):
        return len(self.members)

    def tobytes(self):
        return numpy.asarray(self).tobytes()


class MultiNestedSampler(object):
    def __init__(self, priortransform, multi_loglikelihood, superset_draw_constrained,
                 individual_draw_constrained, draw_constrained, ndata, ndim, nlive_points=200,
                 draw_global_uniform=None, nsuperset_draws=10, use_graph=False,
                 multi_loglikelihood_batch=None, joint_state=None, priortransform_batch=None,
                 device_groups=False, native=None):
        self.nlive_points = nlive_points
        self.nsuperset_draws = nsuperset_draws
        self.priortransform = priortransform
        self.real_multi_loglikelihood = multi_loglikelihood
        self.multi_loglikelihood = multi_loglikelihood
        self.real_multi_loglikelihood_batch = multi_loglikelihood_batch
        self.multi_loglikelihood_batch = multi_loglikelihood_batch
        self.superset_draw_constrained = superset_draw_constrained
        self.individual_draw_constrained = individual_draw_constrained
        self.draw_constrained = draw_constrained
        self.global_iter = 0
        self.ndim = ndim
        self.ndata = ndata
        self.use_graph = use_graph
        #: with use_graph: connected components and distinct ids from the device
        #: (massivedatans_amd.grouping, csrc/mdns_groups.hip) instead of the host walk
        self._device_groups_wanted = bool(device_groups)
        self._dgroups = None
        self._last_selection = None         # (mask object, its members, their original indices) of the last grouping
        self.point_data_map = None          # point id -> set of data sets holding it (lazy)
        #: likelihood evaluations = (candidate, data set) pairs actually scored
        self.nevals = 0
        #: optional: an object keeping live_pointsL, the shelves' likelihoods and the thresholds
        #: (massivedatans_amd.jointstate); a constrained draw is then scored AND decided there
        self.joint = joint_state
        #: optional: a constrainer.NativeContext -- constrainers that are NativeConstrainer objects then
        #: make their whole draw in ONE native call (region, proposals, chunks, commit)
        self.native = native
        self.priortransform_batch = priortransform_batch
        self._live_cache = None
        self.ndraw_calls = 0                # constrained draws made (accepted points)
        self.ndraw_chunks = 0               # chunks of candidates handed to the joint state
        self.draw_seconds = 0.0             # wall-clock spent inside the constrainers' draw_constrained

        # nlive prior draws, every data set starts from the same points: all are superpoints
        # (multi_nested_sampler.py:88-103).  RNG: nlive x uniform(0, 1, ndim).
        all_mask = numpy.ones(ndata) == 1
        jitter_sigma = getattr(self.joint, 'jitter_sigma', 0.0) if self.joint is not None else 0.0
        if jitter_sigma > 0:
            # a likelihood that draws noise with every evaluation (musefuse.py:535): the reference
            # alternates  u = uniform(ndim)  and  L = loglikelihood(x)  per initial point (:90-94),
            # so the noise of point i sits between the coordinates of points i and i + 1 in the stream
            us, noise = [], []
            for _ in range(nlive_points):
                us.append(self.draw_global_uniform())
                noise.append(numpy.random.normal(0, jitter_sigma, size=ndata))
            xs = [priortransform(u) for u in us]
            self.joint.init(numpy.array(xs), jitter=numpy.array(noise))
            Ls = None
        elif multi_loglikelihood_batch is None and self.joint is None:
            # one point at a time, like the reference (:90-94): a likelihood that consumes random
            # numbers itself must see them in that order
            us, xs, Ls = [], [], []
            for _ in range(nlive_points):
                us.append(self.draw_global_uniform())
                xs.append(priortransform(us[-1]))
                Ls.append(multi_loglikelihood(xs[-1], data_mask=all_mask))
        else:
            us = [self.draw_global_uniform() for _ in range(nlive_points)]
            xs = [priortransform(u) for u in us]
        if jitter_sigma > 0 or (multi_loglikelihood_batch is None and self.joint is None):
            pass
        elif self.joint is not None:
            self.joint.init(numpy.array(xs))            # the matrix is computed where it stays
            Ls = None
        elif multi_loglikelihood_batch is not None:
            Ls = list(multi_loglikelihood_batch(numpy.array(xs), all_mask))
        else:
            Ls = [multi_loglikelihood(x, data_mask=all_mask) for x in xs]
        self.nevals += nlive_points * ndata
        # every point ever accepted; kept as views of buffers that grow geometrically (a run of
        # 10 000 data sets accepts millions of points: copying the pile per point was 8 % of it)
        self._pile_u = numpy.array(us)
        self._pile_x = numpy.array(xs)
        self.pointpile = self._pile_u
        self.pointpilex = self._pile_x
        self.live_pointsp = numpy.array([[p] * ndata for p in range(nlive_points)])
        self._live_pointsL = numpy.array(Ls) if Ls is not None else None
        self.superpoints = set(range(nlive_points))
        # how many (live slot, data set) cells hold each point id: the distinct live points of
        # ALL data sets are the ids with a positive count (replaces numpy.unique over the matrix)
        self._refcount = numpy.full(nlive_points, ndata, dtype=int)
        self.data_mask_all = numpy.ones(self.ndata) == 1
        self.real_data_mask_all = numpy.ones(self.ndata) == 1
        self.ndraws = nlive_points
        self._shelves = _Shelves(ndata)
        self._real_indices = None
        self._lpT = None                    # live_pointsp transposed (int32), for the native grouping walk
        self._walk_stale = True
        self._walk = None                   # native incremental grouping walk (csrc/host_groups.c)
        self._low = None                    # smallest live likelihoods per data set (_refresh_thresholds)
        self._low_cap = -1
        if self.native is not None:
            # numpy's normal() may have left half a pair of Gaussian deviates cached: it travels
            # with the stream the native constrainers continue
            self.native.sync_gauss_from_numpy()

    def __del__(self):
        try:
            if self._walk:
                _host.lib().mdns_host_walk_destroy(self._walk)
                self._walk = None
            if self._dgroups is not None:
                self._dgroups.close()
                self._dgroups = None
        except Exception:
            pass

    def _running_indices(self):
        """Original index of every running data set."""
        if self._real_indices is None:
            self._real_indices = numpy.where(self.real_data_mask_all)[0]
        return self._real_indices

    def _device_groups(self):
        """The id matrix on the device (created at the first grouping that needs it), or None."""
        if not self._device_groups_wanted:
            return None
        if self._dgroups is None:
            from .grouping import DeviceGroups
            full = numpy.zeros((self.nlive_points, len(self.real_data_mask_all)), dtype=numpy.int32)
            full[:, self._running_indices()] = self.live_pointsp
            self._dgroups = DeviceGroups(full)
        return self._dgroups

    @property
    def shelves(self):
        """The reference's ``self.shelves`` (lists of ``(pj, uj, xj, Lj)``).  With a joint state the
        likelihoods of the waiting points live THERE (the host queues carry the point ids only,
        and NaN where the reference has Lj): asking for them here is refused instead of answered
        with NaNs."""
        waiting = numpy.arange(self._shelves.L.shape[1])[None, :] < self._shelves.n[:, None]
        if self.joint is not None and numpy.isnan(self._shelves.L[waiting]).any():
            raise RuntimeError("the shelves' likelihoods are kept by the joint state (fetch_rows=False): "
                               "ask joint.thresholds() / use fetch_rows=True")
        return self._shelves.as_lists(self.pointpile, self.pointpilex)

    def _check_shelves_in_step(self):
        """MDNS_DEBUG_SHELVES=1: once per iteration, the shelf sizes of the joint state against the host
        queues of point ids (they are kept in step by two different pieces of code)."""
        _, n_dev = self.joint.thresholds()
        running = self._running_indices()
        if not numpy.array_equal(numpy.asarray(n_dev)[running], self._shelves.n):
            bad = numpy.flatnonzero(numpy.asarray(n_dev)[running] != self._shelves.n)
            raise AssertionError("iteration %d: shelf sizes differ between the joint state and the host queues for data sets %s"
                                 % (self.global_iter, running[bad][:10]))

    @property
    def live_pointsL(self):
        """Likelihoods of the live points, [nlive, running data sets].  With a joint state the
        matrix lives there (on the device) and is fetched when somebody asks."""
        if self.joint is None:
            return self._live_pointsL
        if self._live_cache is None:
            self._live_cache = self.joint.live_matrix()
        return self._live_cache

    @live_pointsL.setter
    def live_pointsL(self, value):
        self._live_pointsL = value

    @property
    def Lmax(self):
        """Highest live likelihood per data set (multi_nested_sampler.py:112,532)."""
        return self.live_pointsL.max(axis=0)

    def draw_global_uniform(self):
        return numpy.random.uniform(0, 1, size=self.ndim)

    # ---- bookkeeping helpers -------------------------------------------------------------
    def get_unique_pointsp(self, allpoints):
        idx = numpy.unique(allpoints)
        return self.pointpile[idx], idx

    def prepare(self):
        """Thresholds of this iteration and shelves purged of entries that no longer beat them
        (multi_nested_sampler.py:130-143)."""
        if self.joint is not None:
            Lmins, Lmini, kept = self.joint.prepare()
            if kept is not None:
                self._shelves.purge_kept(kept)
        else:
            L = self.live_pointsL
            Lmins = L.min(axis=0)
            Lmini = L.argmin(axis=0)
            self._shelves.purge(Lmins)
        allp = numpy.flatnonzero(self._refcount[:len(self.pointpile)])
        return self.pointpile[allp], allp, Lmins.min(), Lmins, Lmini

    def cut_down(self, surviving):
        """Drop the data sets that finished (multi_nested_sampler.py:148-173)."""
        dropped = self.live_pointsp[:, ~surviving]
        if dropped.size:
            self._refcount -= numpy.bincount(dropped.ravel(), minlength=len(self._refcount))
        self.live_pointsp = self.live_pointsp[:, surviving]
        self._lpT = None
        if self.joint is None:
            self._live_pointsL = self._live_pointsL[:, surviving]
        self._live_cache = None
        self._shelves.select(surviving)
        self.ndata = surviving.sum()
        self.data_mask_all = numpy.ones(self.ndata) == 1
        # in place: constrainer caches hold a reference to this array
        self.real_data_mask_all[self.real_data_mask_all] = surviving
        self._real_indices = None
        if self.joint is not None:
            self.joint.set_running(numpy.flatnonzero(self.real_data_mask_all))

        def expand(mask):
            full = self.real_data_mask_all.copy()
            full[full] = mask
            return full

        self.multi_loglikelihood = lambda params, mask: self.real_multi_loglikelihood(params, expand(mask))
        if self.real_multi_loglikelihood_batch is not None:
            self.multi_loglikelihood_batch = \
                lambda params, mask: self.real_multi_loglikelihood_batch(params, expand(mask))
        self.point_data_map = None

    def rebuild_map(self):
        if self.point_data_map is None:
            self.point_data_map = defaultdict(set)
            for d in range(self.ndata):
                for p in self.live_pointsp[:, d]:
                    self.point_data_map[p].add(d)

    # ---- grouping data sets that share live points ---------------------------------------
    def _distinct_points(self, selected):
        """Sorted distinct point ids held by the data sets ``selected`` -- ``numpy.unique`` of
        their columns, by counting instead of sorting -- and the holder count of every id."""
        held = numpy.bincount(self.live_pointsp[:, selected].ravel(), minlength=len(self.pointpile))
        return numpy.flatnonzero(held), held

    def _trivial_groups(self, data_mask, allp):
        """The cases where no decomposition is needed (multi_nested_sampler.py:206-235);
        returns (groups or None, allp, holder counts or None)."""
        selected = numpy.where(data_mask)[0]
        if len(selected) == 1:
            return [(data_mask, self.live_pointsp[:, selected[0]])], allp, None
        held = None
        if len(selected) != len(data_mask) or allp is None:
            allp, held = self._distinct_points(selected)
        if len(allp) < 2 * self.nlive_points:
            # fewer than 2 nlive distinct points over several data sets: some are shared
            return [(data_mask, allp)], allp, held
        if len(self.superpoints) > 0:
            return [(data_mask, allp)], allp, held
        return None, allp, held

    def generate_subsets_nograph(self, data_mask, allp):
        """Groups of data sets connected through shared live points, grown from the first
        unhandled data set by walking its live points in discovery order
        (multi_nested_sampler.py:237-266).  Yields (mask, point ids in discovery order).

        The reference visits every listed point and intersects its holders with the data sets
        still to place.  Most visits find nobody: ``held[p]`` counts the holders of p among the
        data sets still to place, and the walk jumps from one point with holders to the next
        (the counts only fall, so a point found empty stays empty).  Late in a run the selection
        is one component of a thousand data sets and tens of thousands of points, of which a
        few hundred bring somebody in."""
        lib = _host_lib()
        if lib is not None:
            for g in self._groups_native(lib, data_mask):
                yield g
            return
        groups, allp, held = self._trivial_groups(data_mask, allp)
        if groups is not None:
            for g in groups:
                yield g
            return
        for g in self._walk_python(data_mask, held):
            yield g

    def _groups_native(self, lib, data_mask):
        """The native walk (csrc/host_groups.c), incremental over the passes of an iteration:
        the id -> holders index is built for the first selection of the iteration and re-used
        while the selections shrink.  The cases that need no decomposition
        (multi_nested_sampler.py:206-235) are decided there from the number of distinct ids."""
        nsel = int(numpy.count_nonzero(data_mask))
        if nsel == 1:
            yield data_mask, self.live_pointsp[:, numpy.flatnonzero(data_mask)[0]]
            return
        if self._lpT is None:
            # the id matrix with one ROW per data set (the walk reads whole data sets), int32;
            # kept up to date by __next__ / cut_down instead of being rebuilt every iteration
            self._lpT = numpy.ascontiguousarray(self.live_pointsp.T, dtype=numpy.int32)
            self._walk_stale = True
        if self._walk_stale:
            if self._walk is None:
                self._walk = lib.mdns_host_walk_create()
                if not self._walk:
                    raise MemoryError("mdns_host_walk_create")
            # ids are rows of the pile: the walk's per-id arrays are as long as the pile (they
            # are only ever touched at the ids a selection holds)
            if lib.mdns_host_walk_reset(self._walk, self._lpT.ctypes.data, self._lpT.shape[1], self._lpT.shape[0],
                                        len(self.pointpile)) != 0:
                raise MemoryError("mdns_host_walk_reset")
            self._walk_stale = False
        lp = self._lpT
        ndata, nlive = lp.shape
        mask8 = numpy.ascontiguousarray(data_mask, dtype=numpy.uint8)
        group_of = numpy.empty(ndata, dtype=numpy.int32)
        offsets = numpy.empty(ndata + 1, dtype=numpy.int64)
        ndistinct = ctypes.c_int64(0)
        cap = nsel * nlive + nlive                    # a group's list never exceeds its members' ids
        trivial = 1 if len(self.superpoints) > 0 else 0
        points = numpy.empty(cap, dtype=numpy.int32)
        n = lib.mdns_host_walk_groups(self._walk, mask8.ctypes.data, group_of.ctypes.data, points.ctypes.data, cap,
                                      offsets.ctypes.data, ctypes.byref(ndistinct), trivial, 2 * self.nlive_points)
        if n < 0:
            raise MemoryError("mdns_host_walk_groups (%d)" % n)
        if n == 0:
            # some points are shared by all, or there are few of them: one group, ids ascending
            yield data_mask, points[:ndistinct.value]
        elif n == 1:
            yield data_mask.copy(), points[:offsets[1]]
        else:
            for g in range(n):
                yield group_of == g, points[offsets[g]:offsets[g + 1]]

    def _groups_native_stateless(self, lib, data_mask):
        """The same through the stateless entry point (one index build per call); kept as the
        statement the incremental walk is tested against."""
        nsel = int(numpy.count_nonzero(data_mask))
        if nsel == 1:
            yield data_mask, self.live_pointsp[:, numpy.flatnonzero(data_mask)[0]]
            return
        alive = numpy.flatnonzero(self._refcount[:len(self.pointpile)])
        label = numpy.zeros(len(self.pointpile) + 1, dtype=numpy.int64)
        label[alive] = numpy.arange(len(alive))
        lp = numpy.ascontiguousarray(label[self.live_pointsp.T])
        ndata, nlive = lp.shape
        npoints = len(alive)
        mask8 = numpy.ascontiguousarray(data_mask, dtype=numpy.uint8)
        group_of = numpy.empty(ndata, dtype=numpy.int32)
        offsets = numpy.empty(ndata + 1, dtype=numpy.int64)
        cnt = numpy.zeros(npoints, dtype=numpy.int32)
        first = numpy.empty(npoints, dtype=numpy.int64)
        known = numpy.zeros(npoints, dtype=numpy.uint8)
        distinct = numpy.empty(npoints, dtype=numpy.int64)
        ndistinct = ctypes.c_int64(0)
        cap = nsel * nlive + nlive
        want_sorted = 1 if len(self.superpoints) > 0 else 0
        points = numpy.empty(cap, dtype=numpy.int64)
        n = lib.mdns_host_group_walk(lp.ctypes.data, nlive, ndata, mask8.ctypes.data, npoints,
                                     cnt.ctypes.data, first.ctypes.data, known.ctypes.data,
                                     group_of.ctypes.data, points.ctypes.data, cap, offsets.ctypes.data,
                                     distinct.ctypes.data, ctypes.byref(ndistinct), want_sorted,
                                     2 * self.nlive_points)
        if n < 0:
            raise MemoryError("mdns_host_group_walk")
        if ndistinct.value < 2 * self.nlive_points or len(self.superpoints) > 0:
            yield data_mask, alive[distinct[:ndistinct.value]]
        elif n == 1:
            yield data_mask.copy(), alive[points[:offsets[1]]]
        else:
            for g in range(n):
                yield group_of == g, alive[points[offsets[g]:offsets[g + 1]]]

    def _walk_python(self, data_mask, held):
        """The same walk in Python (used when libmdns_host.so is not built, and as the
        statement the native one is tested against)."""
        self.rebuild_map()
        lp = self.live_pointsp
        todo = data_mask.copy()
        todo_set = set(numpy.flatnonzero(todo).tolist())     # same content as `todo`, for set algebra
        if held is None and todo.all():
            held = self._refcount[:len(self.pointpile)].copy()     # every data set: the running id counts
        elif held is None:
            held = numpy.bincount(lp[:, todo].ravel(), minlength=len(self.pointpile))
        known = numpy.zeros(len(held), dtype=bool)
        BLOCK = 2048
        while todo_set:
            first = numpy.where(todo)[0][0]
            todo[first] = False
            todo_set.discard(int(first))
            members = [first]
            column = lp[:, first]
            held[column] -= 1                       # ids within a column are distinct
            points = numpy.empty(max(4 * len(column), 1024), dtype=lp.dtype)
            npoints = len(column)
            points[:npoints] = column
            known[column] = True
            i = 0
            while todo_set:
                # next listed point that a data set still to place holds
                while i < npoints:
                    hit = numpy.flatnonzero(held[points[i:min(npoints, i + BLOCK)]] > 0)
                    if len(hit):
                        i += int(hit[0])
                        break
                    i = min(npoints, i + BLOCK)
                if i >= npoints:
                    break
                # (their order is immaterial: they only enter a mask and a sorted id list)
                newmembers = list(self.point_data_map[int(points[i])] & todo_set)
                members += newmembers
                flat = lp[:, newmembers].ravel()
                if flat.size < 512:
                    numpy.subtract.at(held, flat, 1)
                else:
                    held -= numpy.bincount(flat, minlength=len(held))
                # the reference appends numpy.unique(...) of the new members' live points
                # that are not yet listed: ascending ids, each once (most are listed already)
                unlisted = flat[~known[flat]]
                if unlisted.size:
                    fresh = numpy.unique(unlisted)
                    known[fresh] = True
                    if npoints + len(fresh) > len(points):
                        points = numpy.concatenate((points, numpy.empty(max(len(points), len(fresh)), dtype=points.dtype)))
                    points[npoints:npoints + len(fresh)] = fresh
                    npoints += len(fresh)
                todo[newmembers] = False
                todo_set.difference_update(newmembers)
                i += 1
            known[points[:npoints]] = False            # next component starts clean
            member_mask = numpy.zeros(len(data_mask), dtype=bool)
            member_mask[members] = True
            yield member_mask, points[:npoints].copy()

    def generate_subsets_graph(self, data_mask, allp):
        """Connected components of the bipartite (data set, live point) graph, as the
        reference obtains from igraph (multi_nested_sampler.py:268-355): components in order of
        their lowest data-set index, point ids ascending (igraph numbers the vertices of a
        subgraph in the order they had in the graph -- data sets first, then points by id -- and
        lists a cluster's vertices in that order).  igraph is not available in this image: the ORDER is
        pinned against runs of the reference's own generate_subsets_graph over a stand-in that
        restates igraph's documented numbering (tests/golden/trace_*_graph.npz, oracle/make_trace.py)
        -- i.e. up to igraph's own contract; the PARTITION is that of the pinned walk and is
        cross-checked against networkx in tests/test_sampler_units.py."""
        dg = self._device_groups()
        if dg is not None:
            selected = numpy.flatnonzero(data_mask)
            if len(selected) == 1:
                yield data_mask, self.live_pointsp[:, selected[0]]
                return
            running = self._running_indices()
            real_rows = running[selected]
            rows = None if len(selected) == dg.ndata else real_rows
            ncomp, ids = dg.components(rows, len(self.pointpile))
            # The reference's two shortcuts (:283-297) return ONE group without looking at the
            # graph.  "Fewer than 2 nlive distinct ids" does imply one component; "superpoints
            # known" does not: a point enters `superpoints` when it lands on every shelf (:486-488),
            # before it is live anywhere, and leaves only when it dies somewhere -- so the list can
            # be non-empty while the live-id graph is split.  The reference (and the host paths
            # here) then still draw jointly; so must this one.
            if ncomp == 1 or len(ids) < 2 * self.nlive_points or len(self.superpoints) > 0:
                self._last_selection = (data_mask, selected, real_rows)      # for _fill_shelves
                yield data_mask, ids
                return
            labels, of_id = dg.labels_of_ids(len(ids))
            # components in ascending order of their label (= their lowest data set: igraph's cluster
            # order), members and ids ascending inside: two stable sorts instead of one pass over all
            # data sets and ids per component (a selection can fall into dozens of components)
            # (labels are data-set indices: as 16-bit keys numpy's stable sort is a radix sort)
            narrow = numpy.int16 if len(data_mask) < 32768 and dg.ndata < 32768 else labels.dtype
            labels = labels.astype(narrow)
            order = numpy.argsort(labels, kind='stable')
            cuts = numpy.flatnonzero(numpy.diff(labels[order])) + 1
            of_id = of_id.astype(narrow)
            id_order = numpy.argsort(of_id, kind='stable')
            id_cuts = numpy.flatnonzero(numpy.diff(of_id[id_order])) + 1
            members = numpy.split(selected[order], cuts)
            for group_members, group_ids in zip(members, numpy.split(ids[id_order], id_cuts)):
                # (mask built only for callers that ask: a _LazyMask is a mask for numpy.where & co.)
                yield _LazyMask(len(data_mask), group_members), group_ids, group_members
            return
        lib = _host_lib()
        if lib is not None:
            groups = list(self._groups_native(lib, data_mask))
            if len(groups) == 1:
                mask, points = groups[0]
                yield mask, (points if int(numpy.count_nonzero(data_mask)) == 1 else numpy.sort(points))
                return
            for mask, points in groups:
                yield mask, numpy.sort(points)
            return
        groups, allp, _ = self._trivial_groups(data_mask, allp)
        if groups is not None:
            for g in groups:
                yield g
            return
        selected = numpy.where(data_mask)[0]
        parent = {}

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        for d in selected:
            parent[('n', d)] = ('n', d)
        for d in selected:
            for p in self.live_pointsp[:, d]:
                key = ('p', p)
                if key not in parent:
                    parent[key] = key
                ra, rb = find(('n', d)), find(key)
                if ra != rb:
                    parent[rb] = ra
        comps = defaultdict(lambda: ([], set()))
        for d in selected:
            comps[find(('n', d))][0].append(d)
        for key in list(parent):
            if key[0] == 'p':
                comps[find(key)][1].add(key[1])
        ordered = sorted(comps.values(), key=lambda c: min(c[0]))
        if len(ordered) == 1:
            yield data_mask, allp
            return
        for dsets, points in ordered:
            member_mask = numpy.zeros(len(data_mask), dtype=bool)
            member_mask[dsets] = True
            yield member_mask, numpy.array(sorted(points))

    # ---- one nested-sampling iteration ---------------------------------------------------
    def _refresh_thresholds(self, rows):
        """A data set that already has n accepted points waiting needs the (n+1)-th worst of
        (live points + shelf) as its threshold (multi_nested_sampler.py:438-447).  The reference
        recomputes this for every data set of every draw; live likelihoods do not change within
        an iteration and a shelf only changes when it receives a point, so the thresholds are
        kept in ``self._higher`` and refreshed for exactly the rows whose shelf grew."""
        counts = self._shelves.n[rows]
        waiting = counts > 0
        if not waiting.any():
            return
        rows, counts = rows[waiting], counts[waiting]
        # The n+1 smallest of (live + shelf) lie among the n+1 smallest live values and the
        # shelf, and n never exceeds the shelf capacity: so the cap+1 smallest live likelihoods
        # of every data set, found once per iteration, replace the full columns (a selection,
        # no arithmetic: the thresholds are the same numbers).
        cap = self._shelves.p.shape[1]
        if self._low is None or self._low_cap != cap:
            k = min(cap, self.nlive_points - 1)
            smallest = numpy.partition(self.live_pointsL, k, axis=0)[:k + 1]
            self._low = numpy.ascontiguousarray(numpy.sort(smallest, axis=0).T)     # [ndata, k+1]
            self._low_cap = cap
        # rows in buckets of similar n, so that the arrays sorted are only 2 n + 1 wide
        lo, hi = 0, 1
        while lo < cap:
            pick = (counts > lo) & (counts <= hi)
            if pick.any():
                r, n = rows[pick], counts[pick]
                w = min(hi, cap)
                merged = numpy.hstack((self._low[r, :w + 1], self._shelves.L[r, :w]))
                merged.sort(axis=1)
                self._higher[r] = merged[numpy.arange(len(r)), n]
            lo, hi = hi, 2 * hi + 1

    def _fill_shelves(self, Lmins, allu, allp):
        superset_groups = None
        passes = 0
        while True:
            passes += 1
            self._last_selection = None
            empty = self._shelves.empty()
            if not empty.any():
                return
            if passes == 1 and self.joint is None:
                # thresholds of this iteration (only needed when something has to be drawn)
                self._higher = Lmins.copy()
                self._low = None
                self._refresh_thresholds(numpy.flatnonzero(self._shelves.n > 0))
            focussed = passes > self.nsuperset_draws
            if focussed:
                data_mask = empty
                points = None                     # found by the grouping itself (one counting pass)
            else:
                data_mask = self.data_mask_all
                points = allp
            if superset_groups is not None and not focussed:
                groups = superset_groups
            elif self.use_graph:
                groups = list(self.generate_subsets_graph(data_mask, points))
            else:
                groups = list(self.generate_subsets_nograph(data_mask, points))
            if not focussed and superset_groups is None:
                superset_groups = groups
            assert len(groups) > 0
            rebuilding_draw = focussed or len(groups) > 1

            for group in groups:
                joint_data_mask, joint_live_pointsp = group[0], group[1]
                # (the grouping may have listed the members of the mask it handed back already)
                known = self._last_selection
                if known is not None and known[0] is joint_data_mask:
                    joint_indices, real_rows = known[1], known[2]
                elif len(group) > 2:
                    joint_indices, real_rows = group[2], None
                else:
                    joint_indices, real_rows = numpy.where(joint_data_mask)[0], None
                njoints = len(joint_indices)
                firstd = joint_indices[0]
                max_draws = 100000 if (njoints == 1 and self._shelves.n[firstd] == 0) else 1000
                if len(groups) > 1 and not focussed and (self._shelves.n[joint_indices] > 0).all():
                    continue                      # this group needs nothing
                Lmins_higher = self._higher[joint_indices].copy() if self.joint is None else None
                real_indices = self._running_indices()
                if njoints == 1:
                    draw = self.individual_draw_constrained(real_indices[firstd], self.global_iter, sampler=self)
                elif rebuilding_draw:
                    if real_rows is None:
                        real_rows = real_indices[joint_indices]
                    draw = self.draw_constrained(real_rows, self.real_data_mask_all,
                                                 joint_live_pointsp, self.global_iter)
                else:
                    draw = self.superset_draw_constrained

                constrainer = getattr(draw, '__self__', None) if self.native is not None else None
                if constrainer is not None and hasattr(constrainer, 'draw_native'):
                    # the whole draw in one native call (include/mdns.h Part 5)
                    if njoints == len(real_indices) == self.joint.ndata:
                        rows = None
                    else:
                        rows = real_rows if real_rows is not None else real_indices[joint_indices]
                        rows = numpy.ascontiguousarray(rows, dtype=numpy.int32)
                    t_draw = time.perf_counter()
                    uj, xj, n, bits = constrainer.draw_native(self._pile_u, joint_live_pointsp, rows, njoints)
                    self.draw_seconds += time.perf_counter() - t_draw
                    beats = numpy.unpackbits(bits[:(njoints + 63) // 64].view(numpy.uint8), bitorder='little')[:njoints].view(numpy.bool_)
                    self.joint.took(rows, beats)
                    self._accept(uj, xj, n, njoints, joint_indices, beats, None)
                    continue

                extra = {}
                if self.joint is not None:
                    # the whole chunk of proposed candidates goes to the joint state, which
                    # answers with the first acceptable one (thresholds and accept test there)
                    if njoints == len(real_indices) == self.joint.ndata:
                        rows = None
                    else:
                        rows = real_rows if real_rows is not None else real_indices[joint_indices]
                    last = {}
                    extra['draw_batch'] = lambda us, hint, rows=rows, last=last: self._draw_batch(us, rows, last, hint)
                t_draw = time.perf_counter()
                uj, xj, Lj, n = draw(
                    Lmins=Lmins_higher, priortransform=self.priortransform,
                    loglikelihood=lambda params, m=joint_data_mask: self.multi_loglikelihood(params, m),
                    ndim=self.ndim, draw_global_uniform=self.draw_global_uniform,
                    live_pointsu=self.pointpile[joint_live_pointsp], max_draws=max_draws,
                    iter=self.global_iter, nlive_points=self.nlive_points, **extra)
                self.draw_seconds += time.perf_counter() - t_draw

                if self.joint is not None:
                    self._accept(uj, xj, n, njoints, joint_indices, last['beats'], Lj)      # decided where the thresholds are
                else:
                    beats = Lj > Lmins_higher
                    self._accept(uj, xj, n, njoints, joint_indices, beats, Lj)
                    self._refresh_thresholds(joint_indices[beats])

    def _accept(self, uj, xj, n, njoints, joint_indices, beats, Lj):
        """A constrained draw delivered a point (multi_nested_sampler.py:474-489): it joins the pile
        and the shelves of the data sets whose threshold it beats."""
        self.ndraws += int(n)
        self.ndraw_calls += 1
        self.nevals += int(n) * njoints
        ppi = len(self.pointpile)
        if ppi == len(self._pile_u):
            room = numpy.empty((max(1024, ppi), self._pile_u.shape[1]))
            self._pile_u = numpy.vstack((self._pile_u, room))
            self._pile_x = numpy.vstack((self._pile_x, room))
        self._pile_u[ppi] = uj
        self._pile_x[ppi] = xj
        self.pointpile = self._pile_u[:ppi + 1]
        self.pointpilex = self._pile_x[:ppi + 1]
        # (with a joint state the likelihoods of the waiting points stay there; the host queues
        # carry them only when the state hands the row over)
        self._shelves.append(joint_indices[beats], ppi, Lj[beats] if Lj is not None else numpy.nan)
        nfilled = int(numpy.count_nonzero(beats))
        if len(self._refcount) <= ppi:
            self._refcount = numpy.concatenate((self._refcount, numpy.zeros(max(1024, ppi), dtype=int)))
        if nfilled == self.ndata:
            self.superpoints.add(ppi)
        log.debug('iteration %d: accepted after %d tries, filled %d shelves', self.global_iter, n, nfilled)

    def _draw_batch(self, us, rows, last, hint):
        """One chunk of proposed unit-cube candidates through the joint state: returns
        (index of the first acceptable one or -1, its physical parameters, its likelihood row,
        how many candidates were looked at)."""
        us = us[:self.joint.chunk_size(len(us), self.joint.ndata if rows is None else len(rows), hint)]
        if self.priortransform_batch is not None:
            xs = self.priortransform_batch(us)
        else:
            xs = numpy.array([self.priortransform(u) for u in us])
        idx, Lrow, beats, nscored = self.joint.draw(xs, rows)
        self.ndraw_chunks += 1
        if idx < 0:
            return -1, None, None, nscored
        last['beats'] = beats
        return idx, xs[idx], Lrow, nscored

    def __next__(self):
        allu, allp, _, Lmins, Lmini = self.prepare()
        self._fill_shelves(Lmins, allu, allp)
        if self.joint is not None and _DEBUG_SHELVES:
            self._check_shelves_in_step()

        # every data set gives up its worst live point and takes the head of its shelf
        self.global_iter += 1
        every = numpy.arange(self.ndata)
        dead = self.live_pointsp[Lmini, every]
        uis = self.pointpile[dead]
        xis = self.pointpilex[dead]
        Lis = Lmins if self.joint is not None else self.live_pointsL[Lmini, every]
        if self.point_data_map is not None:
            for d, pj in enumerate(dead):
                self.point_data_map[pj].remove(d)
        if self.superpoints:
            self.superpoints.difference_update(numpy.unique(dead).tolist())
        newp, newL = self._shelves.pop_heads()
        self.live_pointsp[Lmini, every] = newp
        if self._lpT is not None:
            self._lpT[every, Lmini] = newp
        self._walk_stale = True
        if self._dgroups is not None:
            self._dgroups.replace(self._running_indices(), Lmini, newp)
        if self.joint is not None:
            self.joint.advance()
            self._live_cache = None
        else:
            self._live_pointsL[Lmini, every] = newL
        self._refcount -= numpy.bincount(dead, minlength=len(self._refcount))
        self._refcount += numpy.bincount(newp, minlength=len(self._refcount))
        if self.point_data_map is not None:
            for d, pj in enumerate(newp):
                self.point_data_map[pj].add(d)
        return numpy.asarray(uis), numpy.asarray(xis), numpy.asarray(Lis)

    next = __next__

    def __iter__(self):
        while True:
            yield self.__next__()

    def remainder_arrays(self, d):
        """``remainder(d)`` as arrays: (u[nlive, ndim], x[nlive, ndim], L[nlive]) in order of
        increasing likelihood."""
        order = numpy.argsort(self.live_pointsL[:, d])
        p = self.live_pointsp[order, d]
        return self.pointpile[p], self.pointpilex[p], self.live_pointsL[order, d]

    def remainder_arrays_many(self, ds):
        """``remainder_arrays`` of several running data sets at once: (u[nlive, n, ndim], x[nlive, n, ndim],
        L[nlive, n]), every column in order of increasing likelihood (argsort column by column, as there)."""
        ds = numpy.asarray(ds, dtype=int)
        L = numpy.asarray(self.live_pointsL)[:, ds]
        order = numpy.argsort(L, axis=0)
        cols = numpy.arange(len(ds))[None, :]
        p = numpy.asarray(self.live_pointsp)[:, ds][order, cols]
        return self.pointpile[p], self.pointpilex[p], L[order, cols]

    def remainder_likelihoods(self):
        """The likelihoods of ``remainder()`` alone: [nlive, running data sets], every column ascending.
        C-contiguous like the array the integrator used to build from the rows: numpy adds up an
        axis in another order when it is the contiguous one, and the evidence errors are sums."""
        return numpy.ascontiguousarray(numpy.sort(numpy.asarray(self.live_pointsL), axis=0))

    def remainder(self, d=None):
        """Live points in order of increasing likelihood: per data set ``d``, or for all data
        sets at once as (u[ndata,ndim], x, L[ndata]) triples (multi_nested_sampler.py:536-563)."""
        if d is None:
            order = numpy.argsort(self.live_pointsL, axis=0)       # column-wise, like the reference's loop
            every = numpy.arange(self.ndata)
            for i in range(self.nlive_points):
                j = order[i, every]
                p = self.live_pointsp[j, every]
                yield self.pointpile[p], self.pointpilex[p], self.live_pointsL[j, every]
        else:
            for i in numpy.argsort(self.live_pointsL[:, d]):
                p = self.live_pointsp[i, d]
                yield self.pointpile[p], self.pointpilex[p], self.live_pointsL[i, d]


__all__ = ['MultiNestedSampler', 'find_nsmallest']
