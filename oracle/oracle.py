"""ctypes front-end of the CPU oracle.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; nothing under ``massivedatans_amd/`` does.

``Oracle(kind="port")``       -> oracle/liboracle.so   (our C restatement, mdns_oracle.c)
``Oracle(kind="port-omp")``   -> oracle/liboracle-omp.so
``Oracle(kind="reference")``  -> oracle/_ref/*.so  (the reference's own C, compiled by
                                 oracle/Makefile from /root/reference; same symbols as the
                                 reference ctypes bindings: sample.py:85-96, musefuse.py:509-517,
                                 clustering/neighbors.py:100-166)

All three expose the same Python methods, taking the arrays in the REFERENCE layouts
(``yy`` is ``[nx, ndata]`` C-order, ``chosen`` is ``[nsamples, nbootstraps]`` float64).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_bp = np.ctypeslib.ndpointer(dtype=np.bool_, flags="C_CONTIGUOUS")


def build(quiet=True):
    """(Re)build liboracle*.so and, when /root/reference exists, oracle/_ref/*.so."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def have_reference():
    return os.path.exists(os.path.join(HERE, "_ref", "clike.so"))


class Oracle(object):
    def __init__(self, kind="port"):
        self.kind = kind
        if kind in ("port", "port-omp"):
            name = "liboracle.so" if kind == "port" else "liboracle-omp.so"
            path = os.path.join(HERE, name)
            if kind == "port" and os.environ.get("MDNS_ORACLE_LIB"):      # e.g. a sanitizer build (tests/test_sanitizers.py)
                path = os.environ["MDNS_ORACLE_LIB"]
            elif not os.path.exists(path):
                build()
            lib = C.CDLL(path)
            self._gauss = lib.orc_gauss_like
            self._muse = lib.orc_muse_like
            self._nn = lib.orc_nn_maxdist
            self._any = lib.orc_any_within
            self._count = lib.orc_count_within
            self._boot = lib.orc_bootstrap_maxdist
        elif kind in ("reference", "reference-omp"):
            suffix = "-parallel" if kind == "reference-omp" else ""
            ref = os.path.join(HERE, "_ref")
            # the reference never loads clike-parallel (sample.py:81): K1 is always serial
            lk = C.CDLL(os.path.join(ref, "clike.so"))
            lm = C.CDLL(os.path.join(ref, "cmuselike%s.so" % suffix))
            ln = C.CDLL(os.path.join(ref, "cneighbors%s.so" % suffix))
            self._gauss = lk.like
            self._muse = lm.like
            self._nn = ln.most_distant_nearest_neighbor
            self._any = ln.is_within_distance_of
            self._count = ln.count_within_distance_of
            self._boot = ln.bootstrapped_maxdistance
        else:
            raise ValueError(kind)
        self._gauss.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                C.c_double, _bp, _dp]
        self._muse.argtypes = [_dp, _dp, _dp, _bp, C.c_int, C.c_int, _dp]
        self._nn.argtypes = [_dp, C.c_int, C.c_int]
        self._nn.restype = C.c_double
        self._any.argtypes = [_dp, C.c_int, C.c_int, C.c_double, _dp]
        self._any.restype = C.c_int
        self._count.argtypes = [_dp, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int]
        self._boot.argtypes = [_dp, C.c_int, C.c_int, _dp, C.c_int]
        self._boot.restype = C.c_double

    # K1 -- returns the raw chi^2-like sums (caller applies -0.5, sample.py:108)
    def gauss_like(self, x, yy, A, mu, sig, noise_level, data_mask, Lout=None):
        nx, ndata = yy.shape
        data_mask = np.ascontiguousarray(data_mask, dtype=np.bool_)
        if Lout is None:
            Lout = np.zeros(int(data_mask.sum()))
        self._gauss(x, yy, ndata, nx, A, mu, sig, noise_level, data_mask, Lout)
        return Lout

    # K2 -- writes only the masked entries of Lout (length ndata)
    def muse_like(self, yy, vv, ypred, data_mask, Lout=None):
        nx, ndata = yy.shape
        data_mask = np.ascontiguousarray(data_mask, dtype=np.bool_)
        if Lout is None:
            Lout = np.zeros(ndata)
        self._muse(yy, vv, ypred, data_mask, ndata, nx, Lout)
        return Lout

    def most_distant_nearest_neighbor(self, xx):
        n, d = xx.shape
        return self._nn(xx, n, d)

    def is_within_distance_of(self, xx, maxdistance, y):
        n, d = xx.shape
        return self._any(xx, n, d, maxdistance, y) == 1

    def count_within_distance_of(self, xx, maxdistance, yy, countmax=0, out=None):
        n, d = xx.shape
        if out is None:
            out = np.zeros(len(yy))
        self._count(xx, n, d, maxdistance, yy, len(yy), out, countmax)
        return out

    def bootstrapped_maxdistance(self, xx, chosen):
        n, d = xx.shape
        assert chosen.shape[0] == n
        return self._boot(xx, n, d, chosen, chosen.shape[1])
