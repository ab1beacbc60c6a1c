"""Keeps recently used constrainers (regions are expensive to rebuild when the likelihood is
cheap).  Host-side mirror of the reference's ``cachedconstrainer.py:19-116``: a four-generation
cache keyed by the tuple of data-set indices, the "similar to the previous call" shortcut, and
one long-lived constrainer per individual data set.
"""
import logging

import numpy

from .hiermetriclearn import MetricLearningFriendsConstrainer

log = logging.getLogger("massivedatans_amd")


def generate_fresh_constrainer_mlfriends():
    """MLFriends with the reference driver's settings (sample.py:133-137,
    cachedconstrainer.py:8-12)."""
    return MetricLearningFriendsConstrainer(
        metriclearner='truncatedscaling', force_shrink=True,
        rebuild_every=1000, metric_rebuild_every=20, verbose=False)


#: factory used for every new constrainer; the driver may replace it (sample.py:157)
generate_fresh_constrainer = generate_fresh_constrainer_mlfriends


class CachedConstrainer(object):
    """``get(mask, realmask, points, it)`` returns the ``draw_constrained`` of a constrainer
    for the data sets ``mask`` (indices), re-using one from this or the previous three
    nested-sampling iterations when the same index tuple was seen."""

    def __init__(self, sampler=None):
        self.iter = -1
        self.generations = [{}, {}, {}, {}]        # current, previous, ..., oldest
        self.last_mask = []
        self.last_points = []
        self.last_realmask = None
        self.last_key = None
        self._last_table = None
        self.sampler = sampler

    # names kept for readers of the reference
    @property
    def curr_generation(self):
        return self.generations[0]

    def _advance_to(self, it):
        while self.iter < it:
            self.generations = [{}] + self.generations[:3]
            self.last_mask = []
            self.last_realmask = None
            self.last_points = []
            self.last_key = None
            self._last_table = None
            self.iter += 1

    def _similar_to_last(self, mask, realmask, points):
        """The call just before used a slightly larger set of data sets and live points
        (cachedconstrainer.py:54-62): not worth a new region."""
        if self.last_realmask is None:
            return False
        return (len(mask) < len(self.last_mask) and len(mask) > 0.80 * len(self.last_mask)
                and len(points) <= len(self.last_points) and len(points) > 0.90 * len(self.last_points)
                and (self.last_realmask is realmask or numpy.mean(self.last_realmask == realmask) > 0.80)
                and self._all_among_last(points))

    def _all_among_last(self, points):
        """``numpy.isin(points, self.last_points).all()`` through a table of the point ids
        (they are rows of the pile: small non-negative integers), built once per ``last_points``."""
        points = numpy.asarray(points)
        if self._last_table is None:
            last = numpy.asarray(self.last_points)
            self._last_table = numpy.zeros(int(last.max()) + 1 if len(last) else 1, dtype=bool)
            self._last_table[last] = True
        if len(points) and int(points.max()) >= len(self._last_table):
            return False
        return bool(self._last_table[points].all())

    def get(self, mask, realmask, points, it):
        self._advance_to(it)
        if self._similar_to_last(mask, realmask, points):
            return self.generations[0][self.last_key].draw_constrained
        # (the reference keys its dictionaries with tuple(mask): the bytes of the index array
        # identify the same selections at a hundredth of the cost)
        key = (len(mask), numpy.ascontiguousarray(mask, dtype=numpy.int64).tobytes())
        self.last_realmask = realmask
        self.last_mask = mask
        self.last_key = key
        self.last_points = points
        self._last_table = None
        current = self.generations[0]
        if key not in current:
            for older in self.generations[1:]:
                if key in older:
                    current[key] = older[key]
                    break
            else:
                current[key] = generate_fresh_constrainer()
                current[key].sampler = self.sampler
        return current[key].draw_constrained


def generate_individual_constrainer(rebuild_every=1000, metric_rebuild_every=20):
    """One constrainer per data set, forced to rebuild its region when it was last used more
    than five iterations ago (cachedconstrainer.py:92-109)."""
    constrainers = {}
    last_used = {}

    def individual_draw_constrained(i, it, sampler):
        if i not in constrainers:
            constrainers[i] = generate_fresh_constrainer()
            constrainers[i].sampler = sampler
            last_used[i] = it
        if it > last_used[i] + 5:
            constrainers[i].region = None
        last_used[i] = it
        return constrainers[i].draw_constrained

    return constrainers, last_used, individual_draw_constrained


def generate_superset_constrainer():
    return generate_fresh_constrainer()
