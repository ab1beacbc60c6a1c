"""ctypes binding of ``libmdns_host.so`` (csrc/host_groups.c, csrc/host_rng.c): plain C helpers
of the HOST orchestration -- integer graph work and the two scalar-heavy RNG / pow spots.  No
GPU code.  Optional: every user has a Python statement of the same thing to fall back on (and
to be tested against)."""
import ctypes
import os

import numpy

_LIB = None


def lib():
    """The library handle, or None when it has not been built."""
    global _LIB
    if _LIB is None:
        # (MDNS_HOST_LIB: another build of the same sources, e.g. the sanitizer build of tests/test_sanitizers.py)
        path = os.environ.get("MDNS_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmdns_host.so")
        try:
            L = ctypes.CDLL(path)
            L.mdns_host_group_walk.restype = ctypes.c_int
            L.mdns_host_group_walk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]
            L.mdns_host_walk_create.restype = ctypes.c_void_p
            L.mdns_host_walk_create.argtypes = []
            L.mdns_host_walk_destroy.restype = None
            L.mdns_host_walk_destroy.argtypes = [ctypes.c_void_p]
            L.mdns_host_walk_reset.restype = ctypes.c_int
            L.mdns_host_walk_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
            L.mdns_host_walk_groups.restype = ctypes.c_int
            L.mdns_host_walk_groups.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int64]
            L.mdns_host_bootstrap_masks.restype = ctypes.c_int
            L.mdns_host_bootstrap_masks.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
            L.mdns_host_bootstrap_masks_mt.restype = ctypes.c_int
            L.mdns_host_bootstrap_masks_mt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
            L.mdns_host_minmax.restype = None
            L.mdns_host_minmax.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            L.mdns_host_pow10.restype = None
            L.mdns_host_pow10.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
            _LIB = L
        except (OSError, AttributeError):
            _LIB = False
    return _LIB or None


_BITGEN = None


def _global_bitgen():
    """Address of the ``bitgen_t`` of numpy's GLOBAL legacy RandomState (``numpy.random.seed`` /
    ``set_state`` re-seed it in place, so the address holds for the life of the process)."""
    global _BITGEN
    if _BITGEN is None:
        bit_generator = numpy.random.mtrand._rand._bit_generator
        _BITGEN = (bit_generator, bit_generator.ctypes.bit_generator)      # keep the owner alive
    return _BITGEN[1]


_MT_STATE = None


def _mt_state_address():
    """Address of the Mersenne Twister state behind the global legacy stream, for the in-place
    stepping of csrc/host_rng.c -- or 0 when that does not reproduce numpy's own draws (checked
    once, on the live state, which is put back afterwards)."""
    global _MT_STATE
    if _MT_STATE is None:
        _MT_STATE = 0
        L = lib()
        rs = numpy.random.mtrand._rand
        try:
            if L is not None and type(rs._bit_generator).__name__ == "MT19937":
                address = rs._bit_generator.ctypes.state_address
                saved = rs.get_state()
                ok = True
                for K in (1, 2, 3, 700, 5000):
                    want = numpy.zeros(K, dtype=numpy.uint32)
                    rs.set_state(saved)
                    rs.uniform(size=K % 7)                           # move off the saved position
                    before = rs.get_state()
                    L.mdns_host_bootstrap_masks(_global_bitgen(), K, 10, want.ctypes.data)
                    after = rs.get_state()
                    rs.set_state(before)
                    got = numpy.zeros(K, dtype=numpy.uint32)
                    ok = ok and L.mdns_host_bootstrap_masks_mt(address, K, 10, got.ctypes.data) == 0
                    mine = rs.get_state()
                    ok = ok and bool((got == want).all()) and mine[2] == after[2] and bool((mine[1] == after[1]).all())
                rs.set_state(saved)
                if ok:
                    _MT_STATE = address
        except Exception:       # noqa: BLE001 -- any surprise: stay with numpy's own interface
            _MT_STATE = 0
    return _MT_STATE


def bootstrap_masks(nsamples, nbootstraps):
    """Packed bootstrap choice from the global legacy stream (see csrc/host_rng.c), or None when
    the native helper is unavailable."""
    L = lib()
    if L is None:
        return None
    masks = numpy.zeros(nsamples, dtype=numpy.uint32)
    mt = _mt_state_address()
    if mt:
        if L.mdns_host_bootstrap_masks_mt(mt, nsamples, nbootstraps, masks.ctypes.data) == 0:
            return masks
        masks[:] = 0
    if L.mdns_host_bootstrap_masks(_global_bitgen(), nsamples, nbootstraps, masks.ctypes.data) != 0:
        return None
    return masks


def minmax(points):
    """``(points.min(axis=0), points.max(axis=0))`` of a C-contiguous f64[n, ndim] array."""
    L = lib()
    if L is None or len(points) == 0:
        return numpy.min(points, axis=0), numpy.max(points, axis=0)
    points = numpy.ascontiguousarray(points, dtype=numpy.float64)
    lo = numpy.empty(points.shape[1])
    hi = numpy.empty(points.shape[1])
    L.mdns_host_minmax(points.ctypes.data, points.shape[0], points.shape[1], lo.ctypes.data, hi.ctypes.data)
    return lo, hi


def pow10(values):
    """``[10 ** v for v in values]`` as an array (the C library's pow, as the scalar ``**``)."""
    values = numpy.ascontiguousarray(values, dtype=numpy.float64)
    L = lib()
    if L is None:
        return numpy.array([10 ** v for v in values.ravel()]).reshape(values.shape)
    out = numpy.empty_like(values)
    L.mdns_host_pow10(values.ctypes.data, values.size, out.ctypes.data)
    return out
