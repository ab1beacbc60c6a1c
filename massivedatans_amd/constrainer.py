"""One native call per constrained draw.

``NativeConstrainer`` is the reference's ``MetricLearningFriendsConstrainer``
(hiermetriclearn.py:27-211) with its whole ``draw_constrained`` -- rebuild policy, region
construction with the bootstrap draws, candidate generators, prior transform, chunked accept
loop -- behind ``mdns_constrainer_draw`` (include/mdns.h Part 5, csrc/host_constrainer.cpp).
Python keeps what the reference keeps outside the constrainer: which constrainer serves which
group of data sets (cachedconstrainer.py) and the sampler's bookkeeping.

The random numbers come from numpy's own global legacy stream (its Mersenne-Twister state is
stepped in place), so Python code before, between and after native draws sees the stream
exactly where the reference would have left it.

``NativeContext`` holds what all constrainers of a sampler share: the table of device entry
points (``hip_backend``: libmdns_hip.so for a ``GaussJointState``; ``python_backend``: any
joint state / member-set implementation in Python -- the CPU oracle in tests), the prior
transform and the two numpy operations the constrainer leaves to numpy.
"""
import ctypes as C
import os

import numpy

from . import _host, _lib

MAX_DIM = 16

_REGION_CREATE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int,
                             C.POINTER(C.c_uint), C.c_int, C.POINTER(C.c_double))
_REGION_DESTROY = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_REGION_COUNT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int))
_DRAW_BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int)
_DRAW_CHUNK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                          C.POINTER(C.c_ulonglong), C.POINTER(C.c_int))
_CHUNK_SIZE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
_REGION_BEGIN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_uint), C.c_int)
_REGION_RADIUS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double))
_DRAW_BAND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int),
                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)
_DRAW_BAND_COMMIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong))
_DRAW_BAND_BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))
_DRAW_BAND_READY = C.CFUNCTYPE(C.c_int, C.c_void_p)
_DRAW_BAND_END = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)
_CHAIN_BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_CHAIN_END = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                         C.POINTER(C.c_ulonglong), C.POINTER(C.c_double))
_CUSTOM_PRIOR = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))
_VEC_POW = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_double)
_FIT_METRIC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int,
                          C.POINTER(C.c_double), C.POINTER(C.c_double))


class DrawBackend(C.Structure):            # mdns_draw_backend
    _fields_ = [("user", C.c_void_p), ("region_create", _REGION_CREATE), ("region_destroy", _REGION_DESTROY),
                ("region_count", _REGION_COUNT), ("draw_begin", _DRAW_BEGIN), ("draw_chunk", _DRAW_CHUNK),
                ("chunk_size", _CHUNK_SIZE),
                # optional halves of region_create (NULL: not offered): K6 launched / its radius awaited
                ("region_begin", _REGION_BEGIN), ("region_radius", _REGION_RADIUS),
                # optional: the likelihood noise in band form (mdns.h Part 5)
                ("draw_band", _DRAW_BAND), ("draw_band_commit", _DRAW_BAND_COMMIT),
                # optional: the first batch of a region without a host look in between (mdns.h Part 5)
                ("chain_begin", _CHAIN_BEGIN), ("chain_end", _CHAIN_END),
                # optional: draw_band in two halves, the noise bounds of the next chunk made in between
                ("draw_band_begin", _DRAW_BAND_BEGIN), ("draw_band_ready", _DRAW_BAND_READY), ("draw_band_end", _DRAW_BAND_END)]


class Prior(C.Structure):                  # mdns_prior
    _fields_ = [("ndim", C.c_int), ("nparams", C.c_int), ("a", C.c_double * MAX_DIM), ("b", C.c_double * MAX_DIM),
                ("pow10", C.c_int * MAX_DIM), ("kernel_pow10", C.c_int * MAX_DIM), ("custom", _CUSTOM_PRIOR),
                ("user", C.c_void_p), ("jitter_sigma", C.c_double)]


class NumpyOps(C.Structure):               # mdns_numpy_ops
    _fields_ = [("user", C.c_void_p), ("vec_pow", _VEC_POW), ("fit_metric", _FIT_METRIC)]


METRICS = {'none': 0, 'simplescaling': 1, 'truncatedscaling': 2}
#: mdns_constrainer_stats (include/mdns.h)
COUNTERS = ("draws", "chunks", "candidates", "pairs", "regions", "radii", "counts", "proposals", "inside", "tries",
            "ns_bootstrap", "ns_region", "ns_count", "ns_propose", "ns_transform", "ns_chunk", "ns_draw", "ns_jitter",
            "chains", "chain_counts", "param_mismatch", "ns_chain", "band_pairs", "band_replays", "band_ahead")

_HOST = None


def host_lib():
    """libmdns_host.so with the constrainer entry points declared, or None."""
    global _HOST
    if _HOST is None:
        L = _host.lib()
        if L is None or not hasattr(L, "mdns_constrainer_draw"):
            _HOST = False
        else:
            L.mdns_constrainer_create.restype = C.c_void_p
            L.mdns_constrainer_create.argtypes = [C.c_int] * 5
            L.mdns_constrainer_destroy.restype = None
            L.mdns_constrainer_destroy.argtypes = [C.c_void_p, C.c_void_p]
            L.mdns_constrainer_forget_region.restype = None
            L.mdns_constrainer_forget_region.argtypes = [C.c_void_p]
            L.mdns_constrainer_draw.restype = C.c_int
            L.mdns_constrainer_draw.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4
            L.mdns_constrainer_stats.restype = None
            L.mdns_constrainer_stats.argtypes = [C.c_void_p, C.c_void_p]
            L.mdns_constrainer_share_stats.restype = None
            L.mdns_constrainer_share_stats.argtypes = [C.c_void_p, C.c_void_p]
            L.mdns_host_last_error.restype = C.c_char_p
            L.mdns_host_last_error.argtypes = []
            L.mdns_host_rng_get_gauss.restype = None
            L.mdns_host_rng_get_gauss.argtypes = [C.c_void_p, C.c_void_p]
            L.mdns_host_rng_set_gauss.restype = None
            L.mdns_host_rng_set_gauss.argtypes = [C.c_int, C.c_double]
            _HOST = L
    return _HOST or None


def available():
    """The native constrainer can run: library built and numpy's global stream is an MT19937 whose
    state the in-place stepping reproduces (checked once, massivedatans_amd/_host.py)."""
    return host_lib() is not None and bool(_host._mt_state_address())


def sample_py_prior():
    """priortransform of sample.py:52-58 -- A = 10**(2u - 2), mu = 400u + 400, log10 sig = 2u -- and the
    kernel's (A, mu, sig = 10**log_sig) of sample.py:103."""
    p = Prior()
    p.ndim, p.nparams = 3, 3
    for k, (a, b, p10, k10) in enumerate(((2.0, -2.0, 1, 0), (400.0, 400.0, 0, 0), (2.0, 0.0, 0, 1))):
        p.a[k], p.b[k], p.pow10[k], p.kernel_pow10[k] = a, b, p10, k10
    return p


def custom_prior(ndim, nparams, priortransform_batch, kernel_params):
    """A problem definition kept in Python: ``priortransform_batch(us[B, ndim]) -> xs[B, ndim]`` and
    ``kernel_params(xs) -> params[B, nparams]`` called once per chunk."""
    def call(_user, u_ptr, B, x_ptr, p_ptr):
        us = numpy.ctypeslib.as_array(u_ptr, (B, ndim))
        xs = numpy.asarray(priortransform_batch(us), dtype=float)
        numpy.ctypeslib.as_array(x_ptr, (B, ndim))[:] = xs
        numpy.ctypeslib.as_array(p_ptr, (B, nparams))[:] = kernel_params(xs)
    p = Prior()
    p.ndim, p.nparams = ndim, nparams
    p.custom = _CUSTOM_PRIOR(call)
    p._keep = call
    return p


def _numpy_ops():
    """The two operations whose numbers must be numpy's own (see include/mdns.h, mdns_numpy_ops)."""
    from .clustering.sdml import SimpleScaling, TruncatedScaling

    def vec_pow(_user, ptr, n, exponent):
        a = numpy.ctypeslib.as_array(ptr, (n,))
        a[:] = a ** exponent                                  # radfriendsregion.py:156

    def fit_metric(_user, kind, u_ptr, K, ndim, mean_ptr, scale_ptr):
        try:
            u = numpy.ctypeslib.as_array(u_ptr, (K, ndim))
            metric = SimpleScaling() if kind == 1 else TruncatedScaling()
            metric.fit(u - numpy.mean(u, axis=0))             # hiermetriclearn.py:63-65,70-72
            numpy.ctypeslib.as_array(mean_ptr, (ndim,))[:] = metric.mean
            numpy.ctypeslib.as_array(scale_ptr, (ndim,))[:] = metric.scale
            return 0
        except Exception:       # noqa: BLE001 -- reported by the caller as a failed draw
            return 1

    ops = NumpyOps()
    ops.vec_pow = _VEC_POW(vec_pow)
    ops.fit_metric = _FIT_METRIC(fit_metric)
    ops._keep = (vec_pow, fit_metric)
    return ops


def joint_kind_gauss(joint):
    """The Gaussian-line joint state (the chained first batch scores with the K1 chunk kernels)."""
    return getattr(joint, "nparams", 3) == 3


def hip_backend(joint):
    """The device entry points of libmdns_hip.so for a :class:`jointstate.GaussJointState`."""
    lib = _lib.require_device()
    be = DrawBackend()
    be.user = joint._h
    for field, proto, name in (("region_create", _REGION_CREATE, "mdns_backend_region_create"),
                               ("region_destroy", _REGION_DESTROY, "mdns_backend_region_destroy"),
                               ("region_count", _REGION_COUNT, "mdns_backend_region_count"),
                               ("draw_begin", _DRAW_BEGIN, "mdns_backend_draw_begin"),
                               ("draw_chunk", _DRAW_CHUNK, "mdns_backend_draw_chunk"),
                               ("chunk_size", _CHUNK_SIZE, "mdns_backend_chunk_size"),
                               ("region_begin", _REGION_BEGIN, "mdns_backend_region_begin"),
                               ("region_radius", _REGION_RADIUS, "mdns_backend_region_radius")):
        setattr(be, field, C.cast(getattr(lib, name), proto))
    if os.environ.get("MDNS_JITTER_BAND", "1") != "0" and not joint_kind_gauss(joint):
        be.draw_band = C.cast(lib.mdns_backend_draw_band, _DRAW_BAND)
        be.draw_band_commit = C.cast(lib.mdns_backend_draw_band_commit, _DRAW_BAND_COMMIT)
        if os.environ.get("MDNS_BAND_AHEAD", "1") != "0":
            be.draw_band_begin = C.cast(lib.mdns_backend_draw_band_begin, _DRAW_BAND_BEGIN)
            be.draw_band_ready = C.cast(lib.mdns_backend_draw_band_ready, _DRAW_BAND_READY)
            be.draw_band_end = C.cast(lib.mdns_backend_draw_band_end, _DRAW_BAND_END)
    # (a state of the scale-marginalised likelihood chains K6 -> proposals -> membership counts only: its first chunk
    # needs the noise bounds, which the host makes)
    if os.environ.get("MDNS_CHAIN", "1") != "0":
        be.chain_begin = C.cast(lib.mdns_backend_chain_begin, _CHAIN_BEGIN)
        be.chain_end = C.cast(lib.mdns_backend_chain_end, _CHAIN_END)
    be._keep = joint
    return be


def python_backend(joint, member_set_factory=None):
    """The same table over Python objects: ``joint`` with ``draw(params-less xs?)`` -- any joint state of
    :mod:`massivedatans_amd.jointstate` -- and ``member_set_factory(members) -> object`` with
    ``bootstrap_radius_packed(masks, n)``, ``set_radius(r)``, ``count(points)`` (default:
    ``clustering.neighbors.MemberSet``, resolved at call time so that tests can patch it)."""
    from .clustering import neighbors
    regions = {}
    state = {"rows": None, "M": 0, "next": 1}

    def region_create(_user, members_ptr, K, ndim, packed_ptr, nboot, radius_ptr):
        try:
            members = numpy.ctypeslib.as_array(members_ptr, (K, ndim)).copy()
            factory = member_set_factory or neighbors.MemberSet
            if packed_ptr:
                masks = numpy.ctypeslib.as_array(packed_ptr, (K,)).astype(numpy.uint32)
                if hasattr(factory, "bootstrapped"):
                    ms, r = factory.bootstrapped(members, masks, nboot)
                else:
                    ms = factory(members)
                    r = ms.bootstrap_radius_packed(masks, nboot)
                radius_ptr[0] = r
            else:
                ms = factory(members)
                ms.set_radius(radius_ptr[0])
            key = state["next"]
            state["next"] += 1
            regions[key] = ms
            return key
        except Exception:       # noqa: BLE001
            return None

    radii = {}

    def region_begin(user, members_ptr, K, ndim, packed_ptr, nboot):
        # (nothing runs beside Python here: the two halves only keep the native side's bookkeeping honest)
        radius = (C.c_double * 1)()
        key = region_create(user, members_ptr, K, ndim, packed_ptr, nboot, radius)
        if key is not None:
            radii[key] = radius[0]
        return key

    def region_radius(_user, key, radius_ptr):
        if key not in radii:
            return 1
        radius_ptr[0] = radii.pop(key)
        return 0

    def region_destroy(_user, key):
        radii.pop(key, None)
        ms = regions.pop(key, None)
        if ms is not None and hasattr(ms, "close"):
            ms.close()

    def region_count(_user, key, points_ptr, n, counts_ptr):
        try:
            ms = regions[key]
            points = numpy.ctypeslib.as_array(points_ptr, (n, ms_ndim(ms)))
            numpy.ctypeslib.as_array(counts_ptr, (n,))[:] = ms.count(points)
            return 0
        except Exception:       # noqa: BLE001
            return 1

    def ms_ndim(ms):
        return ms.ndim if hasattr(ms, "ndim") else numpy.shape(ms.members)[1]

    def draw_begin(_user, rows_ptr, M):
        state["rows"] = numpy.ctypeslib.as_array(rows_ptr, (M,)).copy() if rows_ptr else None
        state["M"] = M
        return 0

    def draw_chunk(_user, params_ptr, B, jitter_ptr, accepted_ptr, bits_ptr, nscored_ptr):
        try:
            nparams = joint_nparams(joint)
            params = numpy.ctypeslib.as_array(params_ptr, (B, nparams)).copy()
            if jitter_ptr:
                jitter = numpy.ctypeslib.as_array(jitter_ptr, (B, state["M"])).copy()
                idx, _, beats, nscored = joint.draw_params(params, state["rows"], jitter=jitter)
            else:
                idx, _, beats, nscored = joint.draw_params(params, state["rows"])
            accepted_ptr[0] = idx
            nscored_ptr[0] = nscored
            if idx >= 0:
                M = state["M"]
                words = numpy.zeros((M + 63) // 64, dtype=numpy.uint64)
                packed = numpy.packbits(numpy.asarray(beats, dtype=numpy.uint8), bitorder='little')
                words.view(numpy.uint8)[:len(packed)] = packed
                numpy.ctypeslib.as_array(bits_ptr, (len(words),))[:] = words
            return 0
        except Exception:       # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def joint_nparams(j):
        return getattr(j, "nparams", 3)

    # the likelihood noise in band form over a joint state that keeps the scored block (HostJointState):
    # the numpy statement of mdns_backend_draw_band / _commit
    def draw_band(_user, params_ptr, B, bound_ptr, status_ptr, npairs_ptr, pb_ptr, pk_ptr, pL_ptr, pthr_ptr, cap):
        try:
            params = numpy.ctypeslib.as_array(params_ptr, (B, joint_nparams(joint))).copy()
            bound = numpy.ctypeslib.as_array(bound_ptr, (B,)).copy()
            # (a CPU scorer pays per candidate: scored in growing pieces up to the first candidate that is
            # accepted whatever its noise -- nobody looks at the ones behind it)
            rows_L, pos, piece = [], 0, 1
            while pos < B:
                joint.score_params(params[pos:pos + piece], state["rows"])
                rows_L.append(joint._scored_L)
                thr = joint.higher[joint._scored_rows]
                pos += len(rows_L[-1])
                if (rows_L[-1] > thr[None, :] + (1.01 * bound[pos - len(rows_L[-1]):pos, None]
                                                 + 1e-12 * (numpy.abs(rows_L[-1]) + numpy.abs(thr)[None, :]))).any():
                    break
                piece = min(64, 2 * piece)
            L = numpy.vstack(rows_L)
            if len(L) < B:
                L = numpy.vstack((L, numpy.full((B - len(L), L.shape[1]), -numpy.inf)))
            joint._scored_L = L
            band = 1.01 * bound[:, None] + 1e-12 * (numpy.abs(L) + numpy.abs(thr)[None, :])
            band[~numpy.isfinite(L)] = 0.0
            clear = L > thr[None, :] + band
            maybe = ~clear & (L >= thr[None, :] - band)
            status = numpy.where(clear.any(axis=1), 1, numpy.where(maybe.any(axis=1), 2, 0))
            numpy.ctypeslib.as_array(status_ptr, (B,))[:] = status
            b_idx, k_idx = numpy.nonzero(maybe)
            npairs_ptr[0] = len(b_idx)
            n = min(len(b_idx), cap)
            numpy.ctypeslib.as_array(pb_ptr, (cap,))[:n] = b_idx[:n]
            numpy.ctypeslib.as_array(pk_ptr, (cap,))[:n] = k_idx[:n]
            numpy.ctypeslib.as_array(pL_ptr, (cap,))[:n] = L[b_idx[:n], k_idx[:n]]
            numpy.ctypeslib.as_array(pthr_ptr, (cap,))[:n] = thr[k_idx[:n]]
            return 0
        except Exception:       # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def draw_band_commit(_user, b, row_ptr, bits_ptr):
        try:
            M = state["M"]
            joint._scored_L[b] = joint._scored_L[b] + numpy.ctypeslib.as_array(row_ptr, (M,))
            _, beats = joint.commit(b)
            words = numpy.zeros((M + 63) // 64, dtype=numpy.uint64)
            packed = numpy.packbits(numpy.asarray(beats, dtype=numpy.uint8), bitorder='little')
            words.view(numpy.uint8)[:len(packed)] = packed
            numpy.ctypeslib.as_array(bits_ptr, (len(words),))[:] = words
            return 0
        except Exception:       # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    # the two halves: nothing runs beside Python here, but the native side gets its chance to look ahead
    # (MDNS_BAND_READY_AFTER looks at `ready` before the chunk counts as scored)
    pending = {}

    def draw_band_begin(_user, params_ptr, B, bound_ptr):
        pending["params"] = numpy.ctypeslib.as_array(params_ptr, (B, joint_nparams(joint))).copy()
        pending["bound"] = numpy.ctypeslib.as_array(bound_ptr, (B,)).copy()
        pending["polls"] = int(os.environ.get("MDNS_BAND_READY_AFTER", "3"))
        return 0

    def draw_band_ready(_user):
        pending["polls"] -= 1
        return 1 if pending["polls"] < 0 else 0

    def draw_band_end(user, status_ptr, npairs_ptr, pb_ptr, pk_ptr, pL_ptr, pthr_ptr, cap):
        params, bound = pending.pop("params"), pending.pop("bound")
        return draw_band(user, params.ctypes.data_as(C.POINTER(C.c_double)), len(params), bound.ctypes.data_as(C.POINTER(C.c_double)),
                         status_ptr, npairs_ptr, pb_ptr, pk_ptr, pL_ptr, pthr_ptr, cap)

    def chunk_size(_user, offered, M, hint):
        return int(joint.chunk_size(offered, M, hint))

    be = DrawBackend()
    be.user = None
    be.region_create = _REGION_CREATE(region_create)
    be.region_destroy = _REGION_DESTROY(region_destroy)
    if os.environ.get("MDNS_PYTHON_BACKEND_ASYNC", "1") == "1":
        be.region_begin = _REGION_BEGIN(region_begin)
        be.region_radius = _REGION_RADIUS(region_radius)
    be.region_count = _REGION_COUNT(region_count)
    be.draw_begin = _DRAW_BEGIN(draw_begin)
    be.draw_chunk = _DRAW_CHUNK(draw_chunk)
    be.chunk_size = _CHUNK_SIZE(chunk_size)
    if hasattr(joint, "_threshold") and hasattr(joint, "score_params") and os.environ.get("MDNS_JITTER_BAND", "1") != "0":
        be.draw_band = _DRAW_BAND(draw_band)
        be.draw_band_commit = _DRAW_BAND_COMMIT(draw_band_commit)
        if os.environ.get("MDNS_BAND_AHEAD", "1") != "0":
            be.draw_band_begin = _DRAW_BAND_BEGIN(draw_band_begin)
            be.draw_band_ready = _DRAW_BAND_READY(draw_band_ready)
            be.draw_band_end = _DRAW_BAND_END(draw_band_end)
    be._keep = (region_create, region_destroy, region_count, draw_begin, draw_chunk, chunk_size, regions, joint, draw_band, draw_band_commit,
                draw_band_begin, draw_band_ready, draw_band_end)
    return be


class NativeContext(object):
    """What the constrainers of one sampler share."""

    def __init__(self, backend, prior, ndata):
        self.lib = host_lib()
        if self.lib is None:
            raise RuntimeError("libmdns_host.so lacks the native constrainer (make -C massivedatans_amd/csrc)")
        self.mt = _host._mt_state_address()
        if not self.mt:
            raise RuntimeError("numpy's global random stream cannot be stepped natively (not an MT19937?)")
        self.backend, self.prior, self.ops = backend, prior, _numpy_ops()
        self.ndim = int(prior.ndim)
        self._be, self._prior, self._ops = C.addressof(backend), C.addressof(prior), C.addressof(self.ops)
        self.u = numpy.empty(self.ndim)
        self.x = numpy.empty(self.ndim)
        self.bits = numpy.zeros((int(ndata) + 63) // 64 + 1, dtype=numpy.uint64)
        self.ntries = C.c_longlong(0)
        #: counters summed over all constrainers of this context (see COUNTERS)
        self.totals = numpy.zeros(len(COUNTERS), dtype=numpy.int64)
        self._u, self._x, self._bits, self._ntries = self.u.ctypes.data, self.x.ctypes.data, self.bits.ctypes.data, C.addressof(self.ntries)
        self.sync_gauss_from_numpy()

    # the cached second Gaussian deviate of numpy's legacy generator travels with the stream
    def sync_gauss_from_numpy(self):
        st = numpy.random.get_state(legacy=False)
        self.lib.mdns_host_rng_set_gauss(int(st["has_gauss"]), float(st["gauss"]))

    def sync_gauss_to_numpy(self):
        has, val = C.c_int(0), C.c_double(0)
        self.lib.mdns_host_rng_get_gauss(C.addressof(has), C.addressof(val))
        st = numpy.random.get_state(legacy=False)
        if int(st["has_gauss"]) != has.value or (has.value and float(st["gauss"]) != val.value):
            st["has_gauss"], st["gauss"] = has.value, val.value
            numpy.random.set_state(st)

    def fresh_constrainer(self, **kwargs):
        return NativeConstrainer(self, **kwargs)

    def stats(self):
        return dict(zip(COUNTERS, self.totals.tolist()))


class NativeConstrainer(object):
    """Same constructor arguments and attributes the rest of the host code touches as
    ``hiermetriclearn.MetricLearningFriendsConstrainer``; the draw itself is
    :meth:`draw_native` (the sampler calls it instead of ``draw_constrained(**kwargs)``)."""

    def __init__(self, context, metriclearner, rebuild_every=50, metric_rebuild_every=50, verbose=False,
                 keep_phantom_points=False, optimize_phantom_points=False, force_shrink=False):
        self.context = context
        self.metriclearner = metriclearner
        self.rebuild_every, self.metric_rebuild_every = int(rebuild_every), int(metric_rebuild_every)
        self.force_shrink = force_shrink
        self.sampler = None
        self._lib = context.lib
        self._h = self._lib.mdns_constrainer_create(context.ndim, METRICS[metriclearner], self.rebuild_every,
                                                    self.metric_rebuild_every, 1 if force_shrink else 0)
        if not self._h:
            raise RuntimeError("mdns_constrainer_create: " + self._lib.mdns_host_last_error().decode())
        self._lib.mdns_constrainer_share_stats(self._h, context.totals.ctypes.data)

    # `constrainers[i].region = None` (cachedconstrainer.py:104-105) drops the region
    @property
    def region(self):
        raise AttributeError("the region of a native constrainer lives in the library")

    @region.setter
    def region(self, value):
        if value is not None:
            raise ValueError("only None can be assigned")
        self._lib.mdns_constrainer_forget_region(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mdns_constrainer_destroy(self._h, self.context._be)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

    def stats(self):
        out = (C.c_longlong * len(COUNTERS))()
        self._lib.mdns_constrainer_stats(self._h, out)
        return dict(zip(COUNTERS, list(out)))

    def draw_native(self, pile_u, ids, rows, M):
        """One constrained draw: live points ``pile_u[ids]``, data sets ``rows`` (int32, ascending
        original indices; None = all ``M``).  Returns ``(u, x, tries, fill bits uint64[ceil(M/64)])``;
        u, x and the bits are views of buffers that the next draw overwrites."""
        ctx = self.context
        ids = numpy.asarray(ids)
        if ids.dtype.kind not in 'iu' or ids.dtype.itemsize not in (4, 8) or not ids.flags.c_contiguous:
            ids = numpy.ascontiguousarray(ids, dtype=numpy.int64)
        if pile_u.dtype != numpy.float64 or not pile_u.flags.c_contiguous:
            raise ValueError("pile_u must be a C-contiguous float64 array")
        if len(ids) and (int(ids.min()) < 0 or int(ids.max()) >= len(pile_u)):
            raise IndexError("live-point ids outside the pile of %d points" % len(pile_u))
        if rows is not None and (rows.dtype != numpy.int32 or not rows.flags.c_contiguous):
            rows = numpy.ascontiguousarray(rows, dtype=numpy.int32)
        rc = self._lib.mdns_constrainer_draw(self._h, ctx._be, ctx._prior, ctx._ops, ctx.mt, pile_u.ctypes.data,
                                             ids.ctypes.data, ids.dtype.itemsize, len(ids),
                                             rows.ctypes.data if rows is not None else None, M,
                                             ctx._u, ctx._x, ctx._ntries, ctx._bits)
        if rc != 0:
            raise RuntimeError("mdns_constrainer_draw failed: %s | %s"
                               % (self._lib.mdns_host_last_error().decode(), _lib.last_error() if _lib._lib is not None else ""))
        return ctx.u, ctx.x, ctx.ntries.value, ctx.bits

    def draw_constrained(self, **kwargs):
        raise RuntimeError("a native constrainer is driven through draw_native (MultiNestedSampler does so when "
                           "it is built with native=...)")


__all__ = ['NativeConstrainer', 'NativeContext', 'available', 'hip_backend', 'python_backend', 'sample_py_prior',
           'custom_prior']
