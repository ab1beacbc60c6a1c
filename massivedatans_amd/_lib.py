"""ctypes binding of ``libmdns_hip.so`` (C ABI declared in ``include/mdns.h``).

There is no CPU fallback: :func:`load` raises if the library has not been built
(``make -C massivedatans_amd/csrc`` or ``__graft_entry__.build()``), and every wrapper raises
:class:`MdnsError` when the library reports a failure (e.g. no GPU visible).
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmdns_hip.so")
DROPIN_DIR = os.path.join(PKG_DIR, "dropin")

#: every symbol include/mdns.h declares (tests check that the built library exports them all)
ABI_SYMBOLS = (
    "mdns_init", "mdns_device_count", "mdns_last_error", "mdns_abi_version",
    "mdns_gauss_like", "mdns_muse_like", "mdns_register_spectra", "mdns_unregister_spectra",
    "mdns_most_distant_nearest_neighbor", "mdns_is_within_distance_of",
    "mdns_count_within_distance_of", "mdns_bootstrapped_maxdistance",
    "mdns_spectra_create", "mdns_spectra_destroy", "mdns_spectra_ndata", "mdns_spectra_nx",
    "mdns_gauss_loglike_batch", "mdns_muse_loglike_batch", "mdns_muse3_loglike_batch",
    "mdns_region_create", "mdns_region_create_bootstrapped", "mdns_region_wrap_dev", "mdns_region_destroy",
    "mdns_region_bootstrap_radius", "mdns_region_bootstrap_radius_dev", "mdns_region_bootstrap_radius_packed",
    "mdns_region_bootstrap_radius_async", "mdns_region_set_radius",
    "mdns_region_radius", "mdns_region_count", "mdns_region_count_dev", "mdns_region_count_polled",
    "mdns_dev_alloc", "mdns_dev_free", "mdns_h2d", "mdns_d2h", "mdns_d2d", "mdns_sync", "mdns_set_stream",
    "mdns_event_create", "mdns_event_destroy", "mdns_event_record", "mdns_event_elapsed_ms",
    "mdns_profile", "mdns_profile_every", "mdns_profile_read", "mdns_profile_kernel",
    "mdns_gauss_loglike_batch_dev", "mdns_muse_loglike_batch_dev", "mdns_muse3_loglike_batch_dev",
    "mdns_count_within_dev", "mdns_bootstrap_round_maxsq_dev",
    "mdns_joint_create", "mdns_joint_destroy", "mdns_joint_init_gauss", "mdns_joint_init_muse3", "mdns_joint_set_live",
    "mdns_joint_get_live", "mdns_joint_set_running", "mdns_joint_prepare", "mdns_joint_keep_words",
    "mdns_joint_advance", "mdns_joint_reserve", "mdns_joint_shelf_cap", "mdns_joint_draw_gauss", "mdns_joint_score", "mdns_joint_commit",
    "mdns_joint_get_thresholds", "mdns_joint_score_dev", "mdns_joint_flags_dev", "mdns_joint_commit_dev", "mdns_joint_commit_bits_dev",
    "mdns_joint_result_dev", "mdns_joint_result_bytes", "mdns_joint_fetch", "mdns_joint_prepare_dev", "mdns_joint_advance_dev",
    "mdns_joint_restore_live_dev", "mdns_joint_undo_advance_dev", "mdns_joint_live_dev",
    "mdns_groups_create", "mdns_groups_destroy", "mdns_groups_set_ids", "mdns_groups_get_ids",
    "mdns_groups_replace", "mdns_groups_components", "mdns_groups_labels", "mdns_groups_id_labels", "mdns_groups_mean_rounds",
    "mdns_backend_region_create", "mdns_backend_region_destroy", "mdns_backend_region_count",
    "mdns_backend_draw_begin", "mdns_backend_draw_chunk", "mdns_backend_chunk_size",
    "mdns_backend_region_begin", "mdns_backend_region_radius", "mdns_backend_chain_begin", "mdns_backend_chain_end",
    "mdns_backend_draw_score", "mdns_joint_votes_dev", "mdns_backend_draw_commit", "mdns_get_stream",
    "mdns_backend_draw_band", "mdns_backend_draw_band_commit", "mdns_muse_filter_mode", "mdns_muse_filter_stats", "mdns_muse_filter_dev",
    "mdns_backend_draw_band_begin", "mdns_backend_draw_band_ready", "mdns_backend_draw_band_end",
)

#: the symbols of include/mdns.h Part 5 that live in libmdns_host.so (plain host code, no GPU)
HOST_ABI_SYMBOLS = (
    "mdns_constrainer_create", "mdns_constrainer_destroy", "mdns_constrainer_forget_region",
    "mdns_constrainer_draw", "mdns_constrainer_stats", "mdns_constrainer_share_stats", "mdns_host_last_error",
    "mdns_host_rng_get_gauss", "mdns_host_rng_set_gauss",
    # Part 6 (csrc/host_sampler.cpp)
    "mdns_core_create", "mdns_core_destroy", "mdns_core_last_error", "mdns_core_set_host_edges", "mdns_core_set_incremental",
    "mdns_core_set_initial",
    "mdns_core_purge", "mdns_core_fill", "mdns_core_advance", "mdns_core_cut_down", "mdns_core_npoints",
    "mdns_core_nrunning", "mdns_core_pile_u", "mdns_core_pile_x", "mdns_core_get_ids", "mdns_core_get_shelves",
    "mdns_core_get_superpoints", "mdns_core_stats",
)

#: mdns.h MDNS_JOINT_MAX_BATCH
JOINT_MAX_BATCH = 1024


class MdnsError(RuntimeError):
    pass


_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_f64_or_null = C.c_void_p
_lib = None


def _declare(lib):
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    sig = {
        "mdns_init": (i, [i]),
        "mdns_device_count": (i, []),
        "mdns_last_error": (C.c_char_p, []),
        "mdns_abi_version": (i, []),
        "mdns_gauss_like": (i, [vp, vp, i, i, d, d, d, d, vp, vp]),
        "mdns_muse_like": (i, [vp, vp, vp, vp, i, i, vp]),
        "mdns_register_spectra": (i, [vp, vp, i, i]),
        "mdns_unregister_spectra": (i, [vp]),
        "mdns_most_distant_nearest_neighbor": (d, [vp, i, i]),
        "mdns_is_within_distance_of": (i, [vp, i, i, d, vp]),
        "mdns_count_within_distance_of": (i, [vp, i, i, d, vp, i, vp, i]),
        "mdns_bootstrapped_maxdistance": (d, [vp, i, i, vp, i]),
        "mdns_spectra_create": (vp, [vp, vp, vp, i, i, i]),
        "mdns_spectra_destroy": (None, [vp]),
        "mdns_spectra_ndata": (i, [vp]),
        "mdns_spectra_nx": (i, [vp]),
        "mdns_gauss_loglike_batch": (i, [vp, vp, i, d, vp, i, vp]),
        "mdns_muse_loglike_batch": (i, [vp, vp, i, vp, i, vp]),
        "mdns_muse3_loglike_batch": (i, [vp, vp, i, vp, i, vp]),
        "mdns_region_create": (vp, [vp, i, i]),
        "mdns_region_create_bootstrapped": (vp, [vp, i, i, vp, i, vp]),
        "mdns_region_wrap_dev": (vp, [vp, i, i]),
        "mdns_region_destroy": (None, [vp]),
        "mdns_region_bootstrap_radius": (d, [vp, vp, i]),
        "mdns_region_bootstrap_radius_dev": (d, [vp, vp, i]),
        "mdns_region_bootstrap_radius_packed": (d, [vp, vp, i]),
        "mdns_region_bootstrap_radius_async": (i, [vp, vp, i]),
        "mdns_region_set_radius": (i, [vp, d]),
        "mdns_region_radius": (d, [vp]),
        "mdns_region_count": (i, [vp, vp, i, vp]),
        "mdns_region_count_dev": (i, [vp, vp, i, vp]),
        "mdns_region_count_polled": (i, [vp, vp, i, vp]),
        "mdns_dev_alloc": (vp, [sz]),
        "mdns_dev_free": (None, [vp]),
        "mdns_h2d": (i, [vp, vp, sz]),
        "mdns_d2h": (i, [vp, vp, sz]),
        "mdns_d2d": (i, [vp, vp, sz]),
        "mdns_sync": (i, []),
        "mdns_set_stream": (i, [vp]),
        "mdns_event_create": (vp, []),
        "mdns_event_destroy": (None, [vp]),
        "mdns_event_record": (i, [vp]),
        "mdns_event_elapsed_ms": (d, [vp, vp]),
        "mdns_profile": (i, [i]),
        "mdns_profile_every": (i, [i]),
        "mdns_profile_read": (i, [i, vp, vp]),
        "mdns_profile_kernel": (C.c_char_p, [i]),
        "mdns_gauss_loglike_batch_dev": (i, [vp, vp, i, d, vp, i, vp]),
        "mdns_muse_loglike_batch_dev": (i, [vp, vp, i, vp, i, vp]),
        "mdns_muse3_loglike_batch_dev": (i, [vp, vp, i, vp, i, vp]),
        "mdns_count_within_dev": (i, [vp, i, i, d, vp, i, vp]),
        "mdns_bootstrap_round_maxsq_dev": (i, [vp, i, i, vp, i, vp]),
        "mdns_joint_create": (vp, [vp, i, i]),
        "mdns_joint_destroy": (None, [vp]),
        "mdns_joint_init_gauss": (i, [vp, vp, d]),
        "mdns_joint_init_muse3": (i, [vp, vp, vp]),
        "mdns_joint_set_live": (i, [vp, vp]),
        "mdns_joint_get_live": (i, [vp, vp]),
        "mdns_joint_set_running": (i, [vp, vp, i]),
        "mdns_joint_prepare": (i, [vp, vp, vp, vp]),
        "mdns_joint_keep_words": (i, [vp]),
        "mdns_joint_advance": (i, [vp]),
        "mdns_joint_reserve": (i, [vp, i]),
        "mdns_joint_shelf_cap": (i, [vp]),
        "mdns_joint_draw_gauss": (i, [vp, vp, i, d, vp, i, vp, vp, vp]),
        "mdns_joint_score": (i, [vp, vp, i, d, vp, i]),
        "mdns_joint_commit": (i, [vp, vp, vp, vp]),
        "mdns_joint_get_thresholds": (i, [vp, vp, vp]),
        "mdns_joint_score_dev": (i, [vp, vp, i, d, vp, i]),
        "mdns_joint_flags_dev": (vp, [vp]),
        "mdns_joint_commit_dev": (i, [vp, vp, i]),
        "mdns_joint_commit_bits_dev": (i, [vp, vp, i]),
        "mdns_joint_result_dev": (vp, [vp]),
        "mdns_joint_result_bytes": (sz, [i]),
        "mdns_joint_fetch": (i, [vp, i, vp, vp]),
        "mdns_joint_prepare_dev": (i, [vp]),
        "mdns_joint_advance_dev": (i, [vp]),
        "mdns_joint_restore_live_dev": (i, [vp, vp]),
        "mdns_joint_undo_advance_dev": (i, [vp]),
        "mdns_joint_live_dev": (vp, [vp]),
        "mdns_groups_create": (vp, [i, i]),
        "mdns_groups_destroy": (None, [vp]),
        "mdns_groups_set_ids": (i, [vp, vp]),
        "mdns_groups_get_ids": (i, [vp, vp]),
        "mdns_groups_replace": (i, [vp, vp, vp, vp, i]),
        "mdns_groups_components": (i, [vp, vp, i, C.c_longlong, vp, vp, vp, C.c_longlong, vp]),
        "mdns_groups_labels": (i, [vp, vp, vp]),
        "mdns_groups_id_labels": (i, [vp, vp, vp, C.c_longlong]),
        "mdns_groups_mean_rounds": (d, [vp]),
        "mdns_backend_region_create": (vp, [vp, vp, i, i, vp, i, vp]),
        "mdns_backend_region_destroy": (None, [vp, vp]),
        "mdns_backend_region_count": (i, [vp, vp, vp, i, vp]),
        "mdns_backend_draw_begin": (i, [vp, vp, i]),
        "mdns_backend_draw_chunk": (i, [vp, vp, i, vp, vp, vp, vp]),
        "mdns_backend_chunk_size": (i, [vp, i, i, i]),
        "mdns_backend_region_begin": (vp, [vp, vp, i, i, vp, i]),
        "mdns_backend_region_radius": (i, [vp, vp, vp]),
        "mdns_backend_draw_score": (i, [vp, vp, i, vp]),
        "mdns_joint_votes_dev": (vp, [vp]),
        "mdns_backend_draw_commit": (i, [vp, vp, vp]),
        "mdns_get_stream": (vp, []),
        "mdns_backend_draw_band": (i, [vp, vp, i, vp, vp, vp, vp, vp, vp, vp, i]),
        "mdns_backend_draw_band_commit": (i, [vp, i, vp, vp]),
        "mdns_backend_draw_band_begin": (i, [vp, vp, i, vp]),
        "mdns_backend_draw_band_ready": (i, [vp]),
        "mdns_backend_draw_band_end": (i, [vp, vp, vp, vp, vp, vp, vp, i]),
        "mdns_muse_filter_dev": (i, [vp, vp, i, vp, i, vp, vp, vp]),
        "mdns_muse_filter_mode": (None, [i]),
        "mdns_muse_filter_stats": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def load():
    """Load (once) and return the ctypes handle of libmdns_hip.so."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdnsError(
                "libmdns_hip.so is not built (%s): run `make -C massivedatans_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback."
                % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
    return _lib


def last_error():
    return load().mdns_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise MdnsError("%s failed: %s" % (what, last_error()))


def require_device():
    lib = load()
    if lib.mdns_device_count() <= 0:
        raise MdnsError("no HIP device visible: the massivedatans_amd hot path has no CPU fallback")
    check(lib.mdns_init(-1), "mdns_init")
    return lib


def ptr(a):
    """Raw data pointer (an int; every pointer argument is declared c_void_p) of a C-contiguous
    numpy array that the caller keeps alive."""
    return a.ctypes.data


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
