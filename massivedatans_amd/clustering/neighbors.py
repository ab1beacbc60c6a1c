"""RadFriends neighbourhood functions, same names and argument meaning as the reference's
``clustering/neighbors.py:100-231`` ctypes wrappers, running on the GPU through
``libmdns_hip.so``.

There is no scipy fallback here (the reference silently falls back when its library is
missing, neighbors.py:179-182; this module raises instead).
"""
import ctypes as C

import numpy

from .. import _host, _lib


def _pts(a, name):
    a = _lib.as_f64(a)
    if a.ndim != 2:
        raise ValueError("%s must be [n, ndim]" % name)
    return a


def most_distant_nearest_neighbor(xx):
    """max_i min_{j != i} |x_i - x_j|  (neighbors.py:107-110, cneighbors.c:32-75)."""
    xx = _pts(xx, "xx")
    lib = _lib.require_device()
    r = lib.mdns_most_distant_nearest_neighbor(_lib.ptr(xx), xx.shape[0], xx.shape[1])
    if r != r:
        raise _lib.MdnsError("most_distant_nearest_neighbor failed: " + _lib.last_error())
    return r


def is_within_distance_of(xx, maxdistance, y):
    """True if any member is strictly closer than maxdistance to the point y
    (neighbors.py:121-124)."""
    xx = _pts(xx, "xx")
    y = _lib.as_f64(y)
    lib = _lib.require_device()
    r = lib.mdns_is_within_distance_of(_lib.ptr(xx), xx.shape[0], xx.shape[1], float(maxdistance),
                                       _lib.ptr(y))
    if r < 0:
        raise _lib.MdnsError("is_within_distance_of failed: " + _lib.last_error())
    return r == 1


def _count(xx, maxdistance, yy, countmax):
    xx, yy = _pts(xx, "xx"), _pts(yy, "yy")
    if xx.shape[1] != yy.shape[1]:
        raise ValueError("members and points differ in dimension")
    counts = numpy.zeros(len(yy))
    lib = _lib.require_device()
    _lib.check(lib.mdns_count_within_distance_of(
        _lib.ptr(xx), xx.shape[0], xx.shape[1], float(maxdistance), _lib.ptr(yy), len(yy),
        _lib.ptr(counts), countmax), "count_within_distance_of")
    return counts


def count_within_distance_of(xx, maxdistance, yy):
    """Number of members within maxdistance of each point (neighbors.py:137-147)."""
    return _count(xx, maxdistance, yy, 0).astype(int)


def any_within_distance_of(xx, maxdistance, yy):
    """Whether any member is within maxdistance of each point (neighbors.py:149-159)."""
    return _count(xx, maxdistance, yy, 1) > 0


def draw_bootstrap_choice(nsamples, nbootstraps):
    """The ``chosen`` matrix exactly as neighbors.py:170-174 builds it: one
    ``numpy.random.choice(arange(n), size=n, replace=True)`` per round on the GLOBAL legacy
    RNG stream (the call order is part of the results)."""
    # legacy ``choice(arange(n), size=n)`` is ``randint(0, n, size=n)``, value by value, so all
    # rounds come from one ``randint`` of shape (rounds, n): the same numbers from the same
    # position of the stream (tests/test_sampler_units.py compares with the spelled-out calls)
    idx = numpy.random.randint(0, nsamples, size=(nbootstraps, nsamples))
    chosen = numpy.zeros((nsamples, nbootstraps))
    chosen[idx, numpy.arange(nbootstraps)[:, None]] = 1.
    return chosen


def draw_bootstrap_masks(nsamples, nbootstraps, native=True):
    """The same choice as :func:`draw_bootstrap_choice`, from the same position of the RNG stream,
    packed: bit b of ``masks[i]`` is set when point i is chosen in round b (uint32[nsamples],
    nbootstraps <= 16).  Cheaper to build and to upload than the f64 matrix."""
    if nbootstraps > 16:
        raise ValueError("at most 16 rounds fit the packed form")
    # native: the same draws taken from numpy's own bit generator in C (csrc/host_rng.c); the
    # lines below are the statement it is tested against (tests/test_sampler_units.py)
    if native:
        masks = _host.bootstrap_masks(nsamples, nbootstraps)
        if masks is not None:
            return masks
    idx = numpy.random.randint(0, nsamples, size=(nbootstraps, nsamples))
    masks = numpy.zeros(nsamples, dtype=numpy.uint32)
    for b in range(nbootstraps):
        hit = numpy.zeros(nsamples, dtype=numpy.uint32)
        hit[idx[b]] = 1 << b
        masks |= hit
    return masks


def unpack_bootstrap_masks(masks, nbootstraps):
    """masks -> the reference's f64[nsamples, nbootstraps] matrix."""
    return ((masks[:, None] >> numpy.arange(nbootstraps, dtype=numpy.uint32)[None, :]) & 1).astype(float)


def bootstrapped_maxdistance_chosen(xx, chosen):
    """K6 for a given chosen matrix (cneighbors.c:125-179)."""
    xx = _pts(xx, "xx")
    chosen = _lib.as_f64(chosen)
    if chosen.ndim != 2 or chosen.shape[0] != xx.shape[0]:
        raise ValueError("chosen must be [nsamples, nbootstraps]")
    lib = _lib.require_device()
    r = lib.mdns_bootstrapped_maxdistance(_lib.ptr(xx), xx.shape[0], xx.shape[1], _lib.ptr(chosen),
                                          chosen.shape[1])
    if r != r:
        raise _lib.MdnsError("bootstrapped_maxdistance failed: " + _lib.last_error())
    return r


def bootstrapped_maxdistance(xx, nbootstraps):
    """RadFriends safe radius (neighbors.py:169-177)."""
    xx = _pts(xx, "xx")
    return bootstrapped_maxdistance_chosen(xx, draw_bootstrap_choice(xx.shape[0], nbootstraps))


def nearest_rdistance_guess(u, metric='euclidean'):
    """neighbors.py:185-187 (euclidean only: the scipy branch is not part of the hot path)."""
    assert metric == 'euclidean', metric
    return most_distant_nearest_neighbor(u)


def find_rdistance(u, verbose=False, nbootstraps=15, metric='euclidean'):
    """neighbors.py:229-231: dispatches to the bootstrapped radius."""
    assert metric == 'euclidean', metric
    return bootstrapped_maxdistance(u, nbootstraps)


def bounding_box(points):
    """``(min, max)`` of the points along every axis (radfriendsregion.py:69-70 without the radius)."""
    return _host.minmax(_pts(points, "points"))


class MemberSet(object):
    """The members of one RadFriends region, resident on the device for the region's life
    (``mdns_region_*`` of include/mdns.h): the radius is computed and the many membership
    tests of ``RadFriendsRegion.generate`` run against the same HBM copy."""

    def __init__(self, members, _handle=None):
        self._lib = _lib.require_device()
        members = _pts(members, "members")
        self.nmembers, self.ndim = members.shape
        self._h = _handle or self._lib.mdns_region_create(_lib.ptr(members), self.nmembers, self.ndim)
        if not self._h:
            raise _lib.MdnsError("mdns_region_create failed: " + _lib.last_error())

    @classmethod
    def bootstrapped(cls, members, masks, nbootstraps):
        """Members and packed bootstrap choice uploaded together and K6 run in one call:
        returns ``(member set, radius)``."""
        lib = _lib.require_device()
        members = _pts(members, "members")
        masks = numpy.ascontiguousarray(masks, dtype=numpy.uint32)
        if masks.shape != (len(members),):
            raise ValueError("masks must be uint32[nmembers]")
        radius = C.c_double(0)
        h = lib.mdns_region_create_bootstrapped(_lib.ptr(members), members.shape[0], members.shape[1],
                                                _lib.ptr(masks), int(nbootstraps), C.byref(radius))
        if not h:
            raise _lib.MdnsError("mdns_region_create_bootstrapped failed: " + _lib.last_error())
        return cls(members, _handle=h), radius.value

    def bootstrap_radius(self, chosen):
        """K6 for a given chosen matrix; the result becomes the set's radius."""
        chosen = _lib.as_f64(chosen)
        if chosen.ndim != 2 or chosen.shape[0] != self.nmembers:
            raise ValueError("chosen must be [nmembers, nbootstraps]")
        r = self._lib.mdns_region_bootstrap_radius(self._h, _lib.ptr(chosen), chosen.shape[1])
        if r != r:
            raise _lib.MdnsError("mdns_region_bootstrap_radius failed: " + _lib.last_error())
        return r

    def bootstrap_radius_packed(self, masks, nbootstraps):
        """K6 for a packed choice (:func:`draw_bootstrap_masks`)."""
        masks = numpy.ascontiguousarray(masks, dtype=numpy.uint32)
        if masks.shape != (self.nmembers,):
            raise ValueError("masks must be uint32[nmembers]")
        r = self._lib.mdns_region_bootstrap_radius_packed(self._h, _lib.ptr(masks), int(nbootstraps))
        if r != r:
            raise _lib.MdnsError("mdns_region_bootstrap_radius_packed failed: " + _lib.last_error())
        return r

    def set_radius(self, maxdistance):
        _lib.check(self._lib.mdns_region_set_radius(self._h, float(maxdistance)), "mdns_region_set_radius")

    def count(self, points):
        """Number of members strictly within the radius of each point (int array)."""
        points = _pts(points, "points")
        if points.shape[1] != self.ndim:
            raise ValueError("members and points differ in dimension")
        counts = numpy.zeros(len(points), dtype=numpy.int32)
        if len(points):
            _lib.check(self._lib.mdns_region_count(self._h, _lib.ptr(points), len(points), _lib.ptr(counts)),
                       "mdns_region_count")
        return counts.astype(int)

    def any(self, points):
        return self.count(points) > 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mdns_region_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
