"""Axis-scaling metrics used by the RadFriends constrainer (host side, O(K ndim)).

Same classes, methods and numbers as the reference's ``clustering/sdml.py:25-88``; only
``TruncatedScaling`` is used by the default configuration (sample.py:134).
"""
import numpy


class _Comparable(object):
    def __eq__(self, other):
        # the constrainer asks ``self.metric == IdentityMetric()`` (hiermetriclearn.py:77):
        # attribute-wise comparison, like the reference's ``__dict__`` equality
        return self.__dict__ == other.__dict__

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class IdentityMetric(_Comparable):
    """Leaves coordinates unchanged."""

    def fit(self, x):
        return None

    def transform(self, x):
        return x

    def untransform(self, y):
        return y


class _AxisScaling(_Comparable):
    """w = (x - mean) / scale, per axis."""

    def __init__(self, verbose=False):
        self.verbose = verbose

    def transform(self, x):
        return (x - self.mean) / self.scale

    def untransform(self, y):
        return y * self.scale + self.mean


class SimpleScaling(_AxisScaling):
    """Scale = standard deviation of each axis (sdml.py:44-49)."""

    def fit(self, X, W=None):
        self.mean = numpy.mean(X, axis=0)
        centred = X - self.mean
        self.scale = numpy.std(centred, axis=0)


class TruncatedScaling(_AxisScaling):
    """Standard deviations rounded onto powers of two relative to the widest axis, so that the
    metric does not random-walk between rebuilds (sdml.py:68-82).  Note the direction of the
    reference's rounding: ``scale = 2**int(-log2(std / (1.001 max std)))`` -- the widest axis
    gets 1 and NARROWER axes get LARGER divisors."""

    def fit(self, X, W=None):
        self.mean = numpy.mean(X, axis=0)
        centred = X - self.mean
        spread = numpy.std(centred, axis=0)
        widest = spread.max() * 1.001
        steps = (-numpy.log2(spread / widest)).astype(int)
        self.scale = 2 ** (steps.astype(float))
