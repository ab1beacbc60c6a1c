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
    # Part 5's constrainer is plain host code: libmdns_host.so; everything else: libmdns_hip.so
    host = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libmdns_host.so"))
    missing = [s for s in declared if not hasattr(host if s in _lib.HOST_ABI_SYMBOLS else built, s)]
    assert not missing, missing
    assert sorted(_lib.ABI_SYMBOLS + _lib.HOST_ABI_SYMBOLS) == declared
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
    out, members, points = np.zeros(2), np.zeros((3, 2)), np.zeros((2, 2))       # (alive across the call)
    rc = built.mdns_count_within_distance_of(_lib.ptr(members), 3, 2, 0.1, _lib.ptr(points), 2, _lib.ptr(out), 0)
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


def test_committed_bench_lines_keep_the_contract():
    """The bench lines kept under profiles/ (what `python bench.py [--workload muse]` printed on the
    GPU box) carry every key the driver's contract names, the roofline and cpu_baseline objects
    included, and a fraction that is a fraction."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r02_bench*.json")))
    assert paths
    for path in paths:
        line = json.load(open(path))
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                    "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in line, (path, key)
        assert line["unit"] == "likelihood evals/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
        assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic"
        assert "workload" in line["config"] and "model" not in line["config"]
        r = line["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in r, (path, key)
        assert r["bound"] in ("hbm", "fp64_valu") and 0.0 < r["frac"] < 1.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        # value = units per step / time per step
        assert abs(line["value"] * line["ms_per_step"] * 1e-3 / (line["config"]["candidates_per_step"] * line["config"]["spectra_per_gpu"] * line["n_gpus"]) - 1) < 1e-6
        if "cpu_baseline" in line:
            for key in ("value", "unit", "cores", "kind", "sample"):
                assert key in line["cpu_baseline"], (path, key)
            assert line["cpu_baseline"]["kind"] in ("reference", "port")
