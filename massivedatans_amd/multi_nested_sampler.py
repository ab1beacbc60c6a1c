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
  shelves[d]                             FIFO of accepted (id, u, x, L) waiting to replace the
                                         worst live point of data set d
  superpoints                            ids still shared by ALL data sets

What differs from the reference is mechanical: no printing (``logging`` at DEBUG), the initial
``nlive`` likelihood vectors come from one batched launch, and the constrainers get an
optional batch scorer (``loglikelihood_batch``) so that they can look ahead on the GPU.
"""
import logging
from collections import defaultdict

import numpy

log = logging.getLogger("massivedatans_amd")


def find_nsmallest(n, arr1, arr2):
    """(n+1)-th smallest value of the two arrays together (multi_nested_sampler.py:44-47)."""
    merged = numpy.concatenate((arr1, arr2))
    return numpy.partition(merged, n)[n]


class MultiNestedSampler(object):
    def __init__(self, priortransform, multi_loglikelihood, superset_draw_constrained,
                 individual_draw_constrained, draw_constrained, ndata, ndim, nlive_points=200,
                 draw_global_uniform=None, nsuperset_draws=10, use_graph=False,
                 multi_loglikelihood_batch=None):
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
        self.point_data_map = None          # point id -> set of data sets holding it (lazy)
        #: likelihood evaluations = (candidate, data set) pairs actually scored
        self.nevals = 0

        # nlive prior draws, every data set starts from the same points: all are superpoints
        # (multi_nested_sampler.py:88-103).  RNG: nlive x uniform(0, 1, ndim).
        all_mask = numpy.ones(ndata) == 1
        us = [self.draw_global_uniform() for _ in range(nlive_points)]
        xs = [priortransform(u) for u in us]
        if multi_loglikelihood_batch is not None:
            Ls = list(multi_loglikelihood_batch(numpy.array(xs), all_mask))
        else:
            Ls = [multi_loglikelihood(x, data_mask=all_mask) for x in xs]
        self.nevals += nlive_points * ndata
        self.pointpile = numpy.array(us)
        self.pointpilex = numpy.array(xs)
        self.live_pointsp = numpy.array([[p] * ndata for p in range(nlive_points)])
        self.live_pointsL = numpy.array(Ls)
        self.superpoints = list(range(nlive_points))
        self.Lmax = self.live_pointsL.max(axis=0)
        assert self.Lmax.shape == (ndata,)
        self.data_mask_all = numpy.ones(self.ndata) == 1
        self.real_data_mask_all = numpy.ones(self.ndata) == 1
        self.ndraws = nlive_points
        self.shelves = [[] for _ in range(ndata)]

    def draw_global_uniform(self):
        return numpy.random.uniform(0, 1, size=self.ndim)

    # ---- bookkeeping helpers -------------------------------------------------------------
    def get_unique_pointsp(self, allpoints):
        idx = numpy.unique(allpoints)
        return self.pointpile[idx], idx

    def prepare(self):
        """Thresholds of this iteration and shelves purged of entries that no longer beat them
        (multi_nested_sampler.py:130-143)."""
        L = self.live_pointsL
        Lmins = L.min(axis=0)
        Lmini = L.argmin(axis=0)
        for d in range(self.ndata):
            self.shelves[d] = [entry for entry in self.shelves[d] if entry[3] > Lmins[d]]
        allu, allp = self.get_unique_pointsp(self.live_pointsp)
        return allu, allp, L.min(), Lmins, Lmini

    def cut_down(self, surviving):
        """Drop the data sets that finished (multi_nested_sampler.py:148-173)."""
        self.live_pointsp = self.live_pointsp[:, surviving]
        self.live_pointsL = self.live_pointsL[:, surviving]
        self.shelves = [shelf for keep, shelf in zip(surviving, self.shelves) if keep]
        self.ndata = surviving.sum()
        self.Lmax = self.live_pointsL.max(axis=0)
        self.data_mask_all = numpy.ones(self.ndata) == 1
        # in place: constrainer caches hold a reference to this array
        self.real_data_mask_all[self.real_data_mask_all] = surviving

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
    def _trivial_groups(self, data_mask, allp):
        """The cases where no decomposition is needed (multi_nested_sampler.py:206-235);
        returns (groups or None, allp)."""
        selected = numpy.where(data_mask)[0]
        if len(selected) == 1:
            return [(data_mask, self.live_pointsp[:, selected[0]])], allp
        if len(selected) != len(data_mask):
            allp = numpy.unique(self.live_pointsp[:, selected].flatten())
        if len(allp) < 2 * self.nlive_points:
            # fewer than 2 nlive distinct points over several data sets: some are shared
            return [(data_mask, allp)], allp
        if len(self.superpoints) > 0:
            return [(data_mask, allp)], allp
        return None, allp

    def generate_subsets_nograph(self, data_mask, allp):
        """Groups of data sets connected through shared live points, grown from the first
        unhandled data set by walking its live points in discovery order
        (multi_nested_sampler.py:237-266).  Yields (mask, list of point ids)."""
        groups, allp = self._trivial_groups(data_mask, allp)
        if groups is not None:
            for g in groups:
                yield g
            return
        self.rebuild_map()
        todo = data_mask.copy()
        while todo.any():
            first = numpy.where(todo)[0][0]
            todo[first] = False
            members = [first]
            points = self.live_pointsp[:, first].tolist()
            i = 0
            while i < len(points) and todo.any():
                newmembers = [m for m in self.point_data_map[points[i]] if todo[m]]
                members += newmembers
                for newp in numpy.unique(self.live_pointsp[:, newmembers]):
                    if newp not in points:
                        points.append(newp)
                todo[newmembers] = False
                i += 1
            member_mask = numpy.zeros(len(data_mask), dtype=bool)
            member_mask[members] = True
            yield member_mask, points

    def generate_subsets_graph(self, data_mask, allp):
        """Connected components of the bipartite (data set, live point) graph, as the
        reference obtains from igraph (multi_nested_sampler.py:268-355): components in order of
        their lowest data-set index, point ids ascending.  igraph is not available in this
        image, so this ordering is restated from igraph's documented behaviour and is NOT pinned
        against a reference run (the pinned path is ``use_graph=False``)."""
        groups, allp = self._trivial_groups(data_mask, allp)
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
            yield member_mask, sorted(points)

    # ---- one nested-sampling iteration ---------------------------------------------------
    def _thresholds_with_shelves(self, joint_indices, Lmins):
        """A data set that already has n accepted points waiting needs the (n+1)-th worst of
        (live points + shelf) as its threshold (multi_nested_sampler.py:438-447)."""
        higher = Lmins[joint_indices].copy()
        for j, d in enumerate(joint_indices):
            n = len(self.shelves[d])
            if n:
                higher[j] = find_nsmallest(n, self.live_pointsL[:, d], [entry[3] for entry in self.shelves[d]])
        return higher

    def _fill_shelves(self, Lmins, allu, allp):
        superset_groups = None
        passes = 0
        while True:
            passes += 1
            empty = numpy.array([len(self.shelves[d]) == 0 for d in range(self.ndata)])
            if not empty.any():
                return
            focussed = passes > self.nsuperset_draws
            if focussed:
                data_mask = empty
                _, points = self.get_unique_pointsp(self.live_pointsp[:, data_mask])
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

            for joint_data_mask, joint_live_pointsp in groups:
                joint_indices = numpy.where(joint_data_mask)[0]
                njoints = len(joint_indices)
                firstd = joint_indices[0]
                max_draws = 100000 if (njoints == 1 and len(self.shelves[firstd]) == 0) else 1000
                if len(groups) > 1 and not focussed and all(len(self.shelves[d]) > 0 for d in joint_indices):
                    continue                      # this group needs nothing
                Lmins_higher = self._thresholds_with_shelves(joint_indices, Lmins)
                real_indices = numpy.where(self.real_data_mask_all)[0]
                if njoints == 1:
                    draw = self.individual_draw_constrained(real_indices[firstd], self.global_iter, sampler=self)
                elif rebuilding_draw:
                    draw = self.draw_constrained(real_indices[joint_indices], self.real_data_mask_all,
                                                 joint_live_pointsp, self.global_iter)
                else:
                    draw = self.superset_draw_constrained

                extra = {}
                if self.multi_loglikelihood_batch is not None:
                    extra['loglikelihood_batch'] = \
                        lambda ps, m=joint_data_mask: self.multi_loglikelihood_batch(ps, m)
                    extra['mask_key'] = (self.ndata, joint_data_mask.tobytes())
                uj, xj, Lj, n = draw(
                    Lmins=Lmins_higher, priortransform=self.priortransform,
                    loglikelihood=lambda params, m=joint_data_mask: self.multi_loglikelihood(params, m),
                    ndim=self.ndim, draw_global_uniform=self.draw_global_uniform,
                    live_pointsu=self.pointpile[joint_live_pointsp], max_draws=max_draws,
                    iter=self.global_iter, nlive_points=self.nlive_points, **extra)

                self.ndraws += int(n)
                self.nevals += int(n) * njoints
                ppi = len(self.pointpile)
                self.pointpile = numpy.vstack((self.pointpile, [uj]))
                self.pointpilex = numpy.vstack((self.pointpilex, [xj]))
                nfilled = 0
                for j, d in enumerate(joint_indices):
                    if Lj[j] > Lmins_higher[j]:
                        self.shelves[d].append((ppi, uj, xj, Lj[j]))
                        nfilled += 1
                if nfilled == self.ndata:
                    self.superpoints.append(ppi)
                log.debug('iteration %d: accepted after %d tries, filled %d shelves', self.global_iter, n, nfilled)

    def __next__(self):
        allu, allp, _, Lmins, Lmini = self.prepare()
        self._fill_shelves(Lmins, allu, allp)

        # every data set gives up its worst live point and takes the head of its shelf
        self.global_iter += 1
        every = numpy.arange(self.ndata)
        dead = self.live_pointsp[Lmini, every]
        uis = self.pointpile[dead]
        xis = self.pointpilex[dead]
        Lis = self.live_pointsL[Lmini, every]
        if self.point_data_map is not None:
            for d, pj in enumerate(dead):
                self.point_data_map[pj].remove(d)
        if self.superpoints:
            for pj in numpy.unique(dead):
                if pj in self.superpoints:
                    self.superpoints.remove(pj)
        for d in range(self.ndata):
            pj, _, _, Lj = self.shelves[d].pop(0)
            self.live_pointsp[Lmini[d], d] = pj
            self.live_pointsL[Lmini[d], d] = Lj
            if self.point_data_map is not None:
                self.point_data_map[pj].add(d)
        self.Lmax = self.live_pointsL.max(axis=0)
        assert self.Lmax.shape == (self.ndata,)
        return numpy.asarray(uis), numpy.asarray(xis), numpy.asarray(Lis)

    next = __next__

    def __iter__(self):
        while True:
            yield self.__next__()

    def remainder(self, d=None):
        """Live points in order of increasing likelihood: per data set ``d``, or for all data
        sets at once as (u[ndata,ndim], x, L[ndata]) triples (multi_nested_sampler.py:536-563)."""
        if d is None:
            order = numpy.empty((self.ndata, self.nlive_points), dtype=int)
            for k in range(self.ndata):
                order[k, :] = numpy.argsort(self.live_pointsL[:, k])
            every = numpy.arange(self.ndata)
            for i in range(self.nlive_points):
                j = order[every, i]
                p = self.live_pointsp[j, every]
                yield self.pointpile[p], self.pointpilex[p], self.live_pointsL[j, every]
        else:
            for i in numpy.argsort(self.live_pointsL[:, d]):
                p = self.live_pointsp[i, d]
                yield self.pointpile[p], self.pointpilex[p], self.live_pointsL[i, d]


__all__ = ['MultiNestedSampler', 'find_nsmallest']
