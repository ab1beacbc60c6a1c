"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mdns.h
declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from massivedatans_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mdns.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdns_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree(built):
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    missing = [s for s in declared if not hasattr(built, s)]
    assert not missing, missing
    assert sorted(_lib.ABI_SYMBOLS) == declared
    assert built.mdns_abi_version() == 1


def test_dropin_shims_export_reference_names(built):
    want = {"clike.so": ["like"], "cmuselike.so": ["like"],
            "cneighbors.so": ["most_distant_nearest_neighbor", "is_within_distance_of",
                              "count_within_distance_of", "bootstrapped_maxdistance"]}
    for name, syms in want.items():
        lib = C.CDLL(os.path.join(_lib.DROPIN_DIR, name))
        for s in syms:
            assert hasattr(lib, s), (name, s)


def test_no_cpu_fallback(built):
    if built.mdns_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.MdnsError):
        _lib.require_device()
    from massivedatans_amd.like import GaussLineSpectra
    with pytest.raises(_lib.MdnsError):
        GaussLineSpectra(np.linspace(0, 1, 4), np.zeros((4, 2)))
    from massivedatans_amd.clustering import neighbors
    with pytest.raises(_lib.MdnsError):
        neighbors.count_within_distance_of(np.zeros((3, 2)), 0.1, np.zeros((2, 2)))
    # the raw ABI reports failure instead of computing
    out = np.zeros(2)
    rc = built.mdns_count_within_distance_of(_lib.ptr(np.zeros((3, 2))), 3, 2, 0.1,
                                             _lib.ptr(np.zeros((2, 2))), 2, _lib.ptr(out), 0)
    assert rc != 0 and b"no HIP device" in built.mdns_last_error()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "massivedatans_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "oracle." not in text and "import oracle" not in text, f


def test_host_helper_library_is_built_and_used(built):
    """libmdns_host.so (the native grouping walk) is built by the same Makefile and picked up
    by the sampler -- the trace tests must exercise the native code, not its Python fallback."""
    from massivedatans_amd import multi_nested_sampler as mns
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmdns_host.so")
    assert os.path.exists(path), "run make -C massivedatans_amd/csrc"
    assert hasattr(C.CDLL(path), "mdns_host_group_walk")
    assert mns._host_lib() is not None
