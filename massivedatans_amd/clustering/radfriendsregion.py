"""RadFriends region: the union of balls of one common radius around the live points.

Host-side mirror of the reference's ``clustering/radfriendsregion.py:58-182``: same
constructor, attributes (``members``, ``maxdistance``, ``lo``, ``hi``) and methods; the
membership / counting / radius work goes to the GPU through
:mod:`massivedatans_amd.clustering.neighbors`.

The candidate generator consumes the GLOBAL legacy numpy RNG in exactly the reference's
order, because the sequence of proposed points is part of the
results.
"""
import numpy

from . import neighbors


class RadFriendsRegion(object):
    #: candidates proposed per pass and per proposal kind (radfriendsregion.py:124)
    BATCH = 1000

    def __init__(self, members, maxdistance=None, metric='euclidean', nbootstraps=10, verbose=False):
        assert metric == 'euclidean', metric
        self.members = members
        self.metric = metric
        self.verbose = verbose
        self._set = None                  # members on the device: uploaded when first needed
        self._masks = self._chosen = None
        self._box = None
        self._nbootstraps = nbootstraps
        self._maxdistance = maxdistance
        if maxdistance is None:
            # bootstrapped safe radius (radfriendsregion.py:62-64 -> neighbors.py:170-177):
            # nbootstraps numpy.random.choice calls on the global stream NOW -- the position of
            # these draws in the stream is part of the results -- while K6 itself, which draws
            # nothing, waits until somebody asks for the radius.  (The first region of a new
            # constrainer is only ever asked whether it is larger than "no previous radius",
            # hiermetriclearn.py:53-54, and is replaced: its K6 never runs.)
            if nbootstraps <= 16:
                self._masks = neighbors.draw_bootstrap_masks(len(members), nbootstraps)
            else:
                self._chosen = neighbors.draw_bootstrap_choice(len(members), nbootstraps)

    @property
    def maxdistance(self):
        if self._maxdistance is None:
            if self._masks is not None:
                self._set, self._maxdistance = neighbors.MemberSet.bootstrapped(self.members, self._masks, self._nbootstraps)
            else:
                self._set = neighbors.MemberSet(self.members)
                self._maxdistance = self._set.bootstrap_radius(self._chosen)
            self._masks = self._chosen = None
        return self._maxdistance

    @maxdistance.setter
    def maxdistance(self, value):
        self._maxdistance = value
        self._box = None
        if self._set is not None:
            self._set.set_radius(value)

    def _member_set(self):
        radius = self.maxdistance                        # may create the set on its way
        if self._set is None:
            self._set = neighbors.MemberSet(self.members)
            self._set.set_radius(radius)
        return self._set

    def _bounds(self):
        if self._box is None:
            lo, hi = neighbors.bounding_box(self.members)
            self._box = (lo - self.maxdistance, hi + self.maxdistance)
        return self._box

    @property
    def lo(self):
        return self._bounds()[0]

    @property
    def hi(self):
        return self._bounds()[1]

    def add_members(self, us):
        radius = self.maxdistance
        self.members = numpy.vstack((self.members, us))
        self._set = None
        self._box = None
        self._maxdistance = radius

    # ---- membership ------------------------------------------------------------------
    def count_nearby_members(self, us):
        return self._member_set().count(us)

    def are_inside(self, us):
        return self._member_set().any(us)

    def is_inside(self, u):
        if not ((u >= self.lo).all() and (u <= self.hi).all()):
            return False
        return bool(self._member_set().any(numpy.asarray(u, dtype=float).reshape((1, -1)))[0])

    def are_near_members(self, us):
        """Boolean [nmembers, npoints] proximity matrix (diagnostics; not on the hot path)."""
        us = numpy.asarray(us, dtype=float)
        diff = self.members[:, None, :] - us[None, :, :]
        return numpy.sqrt((diff ** 2).sum(axis=2)) < self.maxdistance

    def get_nearby_member_ids(self, u):
        return numpy.where(self.are_near_members([u]))[0]

    # ---- proposals -------------------------------------------------------------------
    def generate(self, nmax=0):
        """Yield ``(points, ntotal)``: arrays of proposed points inside the region and the
        number of raw proposals spent since the previous yield.  Alternates the reference's two
        proposal kinds (radfriendsregion.py:128-182):

        box   -- uniform in the bounding box, kept where inside any ball;
        ball  -- a random member plus a uniform point of its ball, kept with probability
                 1/(number of balls covering it), which makes the union uniformly sampled.
        """
        N = self.BATCH
        # like the reference, the ball proposals keep the members and radius the generator
        # started with (radfriendsregion.py:118-120)
        members = self.members
        maxdistance = self.maxdistance
        ndim = numpy.shape(members)[1]
        spent = 0          # proposals since the last yield
        proposed = 0       # proposals in total
        while nmax == 0 or proposed < nmax:
            spent += N
            proposed += N
            us = numpy.random.uniform(self.lo, self.hi, size=(N, ndim))
            inside = self.are_inside(us)
            if inside.any():
                yield us[inside, :], spent
                spent = 0

            centres = members[numpy.random.randint(0, len(members), N), :]
            spent += N
            proposed += N
            direction = numpy.random.normal(0, 1, size=(N, ndim))
            direction = direction / ((direction ** 2).sum(axis=1) ** 0.5).reshape((-1, 1))
            # radius density grows as r^(ndim-1): inverse-CDF of a uniform deviate
            radius = maxdistance * numpy.random.uniform(0, 1, size=(N, 1)) ** (1. / ndim)
            us = centres + direction * radius
            nnear = self.count_nearby_members(us)
            coin = numpy.random.uniform(size=len(us))
            with numpy.errstate(divide='ignore'):
                accept = coin < 1. / nnear
            if not accept.any():
                continue
            yield us[accept, :], spent
            spent = 0
