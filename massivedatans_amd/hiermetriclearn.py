"""MLFriends constrained draws: RadFriends region in a learned axis-scaling metric.

Host-side mirror of the reference's ``hiermetriclearn.py:27-211``
(``MetricLearningFriendsConstrainer``): same constructor arguments, rebuild policy, RNG call
order and ``draw_constrained(Lmins, priortransform, loglikelihood, live_pointsu, ndim, **kw)
-> (u, x, L, n_evals)`` contract.  Differences, none of which change the sequence of results:

* Candidates are handed out from an explicit buffer instead of a nested Python generator, so
  that the candidates ALREADY proposed (and only those -- no RNG is consumed early) can be
  handed to a joint state in chunks (``draw_batch`` keyword, ``_draw_chunks``): the chunked loop
  is the Python statement of what ``csrc/host_constrainer.cpp`` does in one native call.
* ``float > None`` (hiermetriclearn.py:53 on the first build) follows Python 2, where the
  reference was written: the comparison is true, so the first build computes the radius twice
  (SURVEY.md appendix A#1).
"""
import logging

import numpy

from .clustering.radfriendsregion import RadFriendsRegion
from .clustering.sdml import IdentityMetric, SimpleScaling, TruncatedScaling

log = logging.getLogger("massivedatans_amd")


class MetricLearningFriendsConstrainer(object):
    #: proposals requested from the region per refill (hiermetriclearn.py:106)
    REGION_BATCH = 10000
    #: probability of an extra unit-cube proposal round after each region refill (:126)
    CUBE_PROBABILITY = 0.1

    def __init__(self, metriclearner, rebuild_every=50, metric_rebuild_every=50, verbose=False,
                 keep_phantom_points=False, optimize_phantom_points=False, force_shrink=False):
        self.metriclearner = metriclearner
        self.rebuild_every = int(rebuild_every)
        self.metric_rebuild_every = int(metric_rebuild_every)
        self.verbose = verbose
        self.force_shrink = force_shrink
        self.metric = IdentityMetric()
        self.region = None
        self.prev_maxdistance = None
        self.last_cluster_points = None
        self.iter_since_metric_rebuild = 0
        self.ndraws_since_rebuild = 0
        self.direct_draws_efficient = True
        self.clusters = None
        self.generator = None
        self._reset_buffer()
        #: number of (candidate, data set) likelihood evaluations requested
        self.nevals_requested = 0

    # ---- region construction ------------------------------------------------------------
    def _fit_metric(self, u):
        """New metric from the live points shifted to their mean (hiermetriclearn.py:63-80);
        returns (metric, whether it differs from the current one)."""
        shifted = u - numpy.mean(u, axis=0)
        if self.metriclearner == 'none':
            return self.metric, False
        if self.metriclearner == 'simplescaling':
            metric = SimpleScaling()
            metric.fit(shifted)
            return metric, True
        if self.metriclearner == 'truncatedscaling':
            metric = TruncatedScaling()
            metric.fit(shifted)
            changed = self.metric == IdentityMetric() or not numpy.all(self.metric.scale == metric.scale)
            return metric, changed
        raise AssertionError(self.metriclearner)

    def _never_grow(self, region, members_old_metric):
        """``force_shrink``: a rebuilt region may not have a larger radius than the one it
        replaces (hiermetriclearn.py:53-54,88-90).  ``prev_maxdistance is None`` counts as
        "smaller than anything", the Python-2 ordering the reference was written under."""
        previous = self.prev_maxdistance
        if self.force_shrink and (previous is None or region.maxdistance > previous):
            return RadFriendsRegion(members=members_old_metric, maxdistance=previous)
        return region

    def cluster(self, u, ndim, keepMetric=False):
        """(Re)build the region around the live points ``u``; with ``keepMetric`` the axis
        scaling stays as it is, otherwise it is re-learned first (hiermetriclearn.py:48-92)."""
        w_old = self.metric.transform(u)
        if keepMetric:
            self.region = self._never_grow(RadFriendsRegion(members=w_old), w_old)
        else:
            new_metric, changed = self._fit_metric(u)
            self.metric = new_metric
            region = RadFriendsRegion(members=self.metric.transform(u))
            # only a region in the SAME metric is comparable with the previous radius
            if not changed and self.prev_maxdistance is not None:
                region = self._never_grow(region, w_old)
            self.region = region
        self.prev_maxdistance = self.region.maxdistance

    def are_inside_cluster(self, points):
        return self.region.are_inside(self.metric.transform(points))

    def is_inside(self, point):
        if not ((point >= 0).all() and (point <= 1).all()):
            return False
        return self.region.is_inside(self.metric.transform(point))

    def rebuild(self, u, ndim, keepMetric=False):
        same = (self.last_cluster_points is not None and len(self.last_cluster_points) == len(u)
                and numpy.all(self.last_cluster_points == u))
        if same:
            return                                  # identical live points: keep region AND generator
        self.cluster(u=u, ndim=ndim, keepMetric=keepMetric)
        self.last_cluster_points = u
        log.debug('maxdistance: %s', self.region.maxdistance)
        self.generator = self.generate(ndim)
        self._reset_buffer()

    # ---- candidate stream ---------------------------------------------------------------
    def generate(self, ndim):
        """Yield ``(us, ntotal)``: arrays of unit-cube candidates inside the region and the
        number of raw proposals behind the FIRST of them (hiermetriclearn.py:104-137)."""
        ntotal = 0
        N = self.REGION_BATCH
        while True:
            if ndim < 40:
                for ws, n in self.region.generate(N):
                    us = self.metric.untransform(ws)
                    assert us.shape[1] == ndim, us.shape
                    ntotal = ntotal + n
                    inside_cube = numpy.logical_and(us < 1, us > 0).all(axis=1)
                    if inside_cube.any():
                        yield us[inside_cube, :], ntotal
                        ntotal = 0
            if numpy.random.uniform() < self.CUBE_PROBABILITY:
                # occasionally propose from the whole unit cube (efficient while the region is big)
                ntotal = ntotal + N
                us = numpy.random.uniform(size=(N, ndim))
                inside = self.region.are_inside(self.metric.transform(us))
                if inside.any():
                    yield us[inside, :], ntotal
                    ntotal = 0

    def _reset_buffer(self):
        self._buf = None          # candidates already proposed, not yet consumed
        self._buf_pos = 0
        self._buf_ntotal = 0

    def _next_candidate(self):
        """Next candidate and the proposal count the reference's generator would report with it
        (the batch total for the first candidate of a batch, 0 for the others)."""
        if self._buf is None or self._buf_pos >= len(self._buf):
            self._buf, self._buf_ntotal = next(self.generator)      # consumes RNG, exactly on demand
            self._buf_pos = 0
        u = self._buf[self._buf_pos]
        ntotal = self._buf_ntotal if self._buf_pos == 0 else 0
        self._buf_pos += 1
        return u, ntotal

    def _draw_chunks(self, draw_batch, live_pointsu, ndim, region_rebuilt, metric_rebuilt):
        """The accept loop of hiermetriclearn.py:181-211 with the candidates handed over in
        chunks: ``draw_batch(us, expected_tries) -> (index of the first acceptable candidate or -1, x, L, number
        of candidates looked at)`` scores a run of ALREADY proposed candidates and takes the
        accept decision where the thresholds are (the GPU).  A chunk never reaches past the
        candidate after which the reference would rebuild its region (hiermetriclearn.py:198-211),
        so regions, RNG draws and results are those of the one-candidate-at-a-time loop."""
        tries = 0
        while True:
            if self._buf is None or self._buf_pos >= len(self._buf):
                self._buf, self._buf_ntotal = next(self.generator)      # consumes RNG, exactly on demand
                self._buf_pos = 0
                if self._buf_ntotal > 100000:
                    self.direct_draws_efficient = False
            room = len(self._buf) - self._buf_pos
            # candidates that may be consumed before a rebuild condition can become true
            if not region_rebuilt:
                room = min(room, max(1, self.rebuild_every - self.ndraws_since_rebuild + 1))
            if not metric_rebuilt:
                room = min(room, max(1, 201 - tries))
            chunk = self._buf[self._buf_pos:self._buf_pos + room]
            # (how many of them are looked at at once is the scorer's choice: it is told how many
            # tries the previous draw of this constrainer needed)
            idx, x, L, nscored = draw_batch(chunk, max(1, getattr(self, '_last_ntoaccept', 1)))
            used = idx + 1 if idx >= 0 else nscored
            assert 0 < used <= room
            tries += used
            self.ndraws_since_rebuild += used
            self._buf_pos += used
            if idx >= 0:
                self._last_ntoaccept = tries
                return chunk[idx], x, L, tries
            if not region_rebuilt and self.ndraws_since_rebuild > self.rebuild_every:
                region_rebuilt = True
                self.rebuild(numpy.asarray(live_pointsu), ndim, keepMetric=True)
                self.ndraws_since_rebuild = 0
            elif not metric_rebuilt and tries > 200:
                metric_rebuilt = True
                self.rebuild(numpy.asarray(live_pointsu), ndim, keepMetric=False)
                self.iter_since_metric_rebuild = 0

    # ---- the draw -----------------------------------------------------------------------
    def _draw_constrained_prepare(self, Lmins, priortransform, loglikelihood, live_pointsu, ndim, **kwargs):
        """Rebuild policy at the start of a draw (hiermetriclearn.py:152-166): a new region
        after ``rebuild_every`` likelihood calls (or when there is none), with a new metric if
        this constrainer has been asked more than ``metric_rebuild_every`` times since the last
        one.  Returns (region rebuilt, metric rebuilt)."""
        region_due = self.region is None or self.ndraws_since_rebuild > self.rebuild_every
        metric_due = self.iter_since_metric_rebuild > self.metric_rebuild_every
        if not region_due:
            assert self.generator is not None
            return False, False
        self.rebuild(numpy.asarray(live_pointsu), ndim, keepMetric=not metric_due)
        self.ndraws_since_rebuild = 0
        if metric_due:
            self.iter_since_metric_rebuild = 0
        assert self.generator is not None
        return True, metric_due

    def draw_constrained(self, Lmins, priortransform, loglikelihood, live_pointsu, ndim, **kwargs):
        """Propose until a candidate beats the threshold of at least one data set
        (hiermetriclearn.py:173-211).  Returns ``(u, x, L, n_likelihood_calls)``."""
        self.iter_since_metric_rebuild += 1
        region_rebuilt, metric_rebuilt = self._draw_constrained_prepare(
            Lmins, priortransform, loglikelihood, live_pointsu, ndim, **kwargs)
        if kwargs.get('draw_batch') is not None:
            return self._draw_chunks(kwargs['draw_batch'], live_pointsu, ndim, region_rebuilt, metric_rebuilt)
        tries = 0
        while True:
            u, nproposed = self._next_candidate()
            assert (u >= 0).all() and (u <= 1).all(), u
            x = priortransform(u)
            L = loglikelihood(x)
            self.nevals_requested += len(L)
            tries += 1
            self.ndraws_since_rebuild += 1
            if nproposed > 100000:
                self.direct_draws_efficient = False
            if numpy.any(L > Lmins):
                self._last_ntoaccept = tries
                return u, x, L, tries
            # a long unsuccessful streak tightens the region -- each kind at most once per draw
            # (hiermetriclearn.py:198-211); the candidate stream restarts from the new region
            if not region_rebuilt and self.ndraws_since_rebuild > self.rebuild_every:
                region_rebuilt = True
                self.rebuild(numpy.asarray(live_pointsu), ndim, keepMetric=True)
                self.ndraws_since_rebuild = 0
            elif not metric_rebuilt and tries > 200:
                metric_rebuilt = True
                self.rebuild(numpy.asarray(live_pointsu), ndim, keepMetric=False)
                self.iter_since_metric_rebuild = 0
