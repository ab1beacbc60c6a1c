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
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmdns_host.so")
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


def bootstrap_masks(nsamples, nbootstraps):
    """Packed bootstrap choice from the global legacy stream (see csrc/host_rng.c), or None when
    the native helper is unavailable."""
    L = lib()
    if L is None:
        return None
    masks = numpy.zeros(nsamples, dtype=numpy.uint32)
    if L.mdns_host_bootstrap_masks(_global_bitgen(), nsamples, nbootstraps, masks.ctypes.data) != 0:
        return None
    return masks


def pow10(values):
    """``[10 ** v for v in values]`` as an array (the C library's pow, as the scalar ``**``)."""
    values = numpy.ascontiguousarray(values, dtype=numpy.float64)
    L = lib()
    if L is None:
        return numpy.array([10 ** v for v in values.ravel()]).reshape(values.shape)
    out = numpy.empty_like(values)
    L.mdns_host_pow10(values.ctypes.data, values.size, out.ctypes.data)
    return out
