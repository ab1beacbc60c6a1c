"""The sampler core's grouping (csrc/host_sampler.cpp, host_sampler_inc.h) on planted id matrices:
the fresh computation against scipy's connected components, and the incremental path of the focussed
passes -- components kept up to date while the selection shrinks -- against the fresh one (the library
compares every pass itself under MDNS_CORE_CHECK_GROUPS=1; here also from outside)."""
import ctypes as C

import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from massivedatans_amd import constrainer, core


def _core(nlive, ndata, use_graph=True):
    """A core with nothing behind its backend table (the grouping does not reach it)."""
    if not core.available():
        pytest.skip("libmdns_host.so not built")
    L = constrainer.host_lib()
    core._declare(L)
    L.mdns_core_debug_set_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]
    L.mdns_core_debug_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    be, prior = constrainer.DrawBackend(), constrainer.sample_py_prior()
    mt = np.zeros(700, dtype=np.uint32)
    h = L.mdns_core_create(nlive, ndata, 3, 10, 1 if use_graph else 0, 2, 1000, 20, 1, C.addressof(be), C.addressof(prior), None,
                           mt.ctypes.data, None, None, None)
    assert h
    return L, h, (be, prior, mt)


def _groups(L, h, sel, nlive, focussed, restart):
    sel = np.ascontiguousarray(sel, dtype=np.int32)
    group_of = np.full(len(sel), -1, dtype=np.int32)
    cap = len(sel) * nlive + nlive
    ids = np.empty(cap, dtype=np.int32)
    offsets = np.zeros(len(sel) + 1, dtype=np.int64)
    n = L.mdns_core_debug_groups(h, sel.ctypes.data, len(sel), int(focussed), int(restart), group_of.ctypes.data, ids.ctypes.data, cap,
                                 offsets.ctypes.data)
    assert n > 0, L.mdns_core_last_error().decode()
    return [(sel[group_of == g].tolist(), ids[offsets[g]:offsets[g + 1]].tolist()) for g in range(n)]


def _scipy_groups(lp, sel, nlive):
    """generate_subsets_graph's statement (multi_nested_sampler.py:268-355): components of the bipartite
    graph, by lowest data set, members and ids ascending; one group when connected or fewer than
    2 nlive distinct ids."""
    sel = np.asarray(sel)
    if len(sel) == 1:
        return [(sel.tolist(), lp[:, sel[0]].tolist())]
    cols = lp[:, sel]
    ids, inv = np.unique(cols, return_inverse=True)
    inv = inv.reshape(cols.shape)
    nd, M = len(ids), len(sel)
    rows = np.repeat(np.arange(M)[None, :], cols.shape[0], axis=0).ravel()
    graph = coo_matrix((np.ones(rows.size), (rows, M + inv.ravel())), shape=(M + nd, M + nd))
    ncomp, labels = connected_components(graph, directed=False)
    if ncomp == 1 or nd < 2 * nlive:
        return [(sel.tolist(), ids.tolist())]
    out = []
    for lab in sorted(set(labels[:M].tolist()), key=lambda v: int(np.flatnonzero(labels[:M] == v)[0])):
        out.append((sel[labels[:M] == lab].tolist(), ids[labels[M:] == lab].tolist()))
    return out


def _planted(rng, ndata, nlive, ncommunities, bridges):
    """Data sets in communities drawing their ids from a community's pool, plus a few shared ids between
    communities (bridges): removing the right data sets splits components."""
    pool = 3 * nlive
    lp = np.empty((nlive, ndata), dtype=np.int64)
    community = rng.randint(0, ncommunities, size=ndata)
    for d in range(ndata):
        lp[:, d] = community[d] * pool + rng.choice(pool, size=nlive, replace=False)
    npoints = ncommunities * pool
    for _ in range(bridges):
        d = rng.randint(ndata)
        other = rng.randint(ncommunities)
        lp[rng.randint(nlive), d] = other * pool + rng.randint(pool)
    # ids within a column must be distinct
    for d in range(ndata):
        col = lp[:, d]
        while len(set(col.tolist())) < nlive:
            seen, fresh = set(), []
            for v in col:
                while v in seen:
                    v = community[d] * pool + rng.randint(pool)
                seen.add(v)
                fresh.append(v)
            col[:] = fresh
    return lp, npoints


@pytest.mark.parametrize("seed", range(6))
def test_fresh_and_incremental_groupings(seed):
    rng = np.random.RandomState(seed)
    ndata, nlive = [(60, 6), (200, 10), (400, 8), (150, 20), (300, 5), (500, 12)][seed]
    lp, npoints = _planted(rng, ndata, nlive, ncommunities=[3, 6, 12, 4, 20, 8][seed], bridges=[2, 6, 20, 3, 30, 10][seed])
    L, h, keep = _core(nlive, ndata)
    L.mdns_core_set_incremental(h, 4000000, 1)               # the library checks every pass itself, too
    ids32 = np.ascontiguousarray(lp, dtype=np.int32)
    assert L.mdns_core_debug_set_ids(h, ids32.ctypes.data, npoints, 0) == 0
    nsplit = 0
    for chain in range(4):
        sel = np.flatnonzero(rng.uniform(size=ndata) < rng.uniform(0.5, 1.0))
        first = True
        while len(sel) > 0:
            want = _scipy_groups(lp, sel, nlive)
            fresh = _groups(L, h, sel, nlive, focussed=False, restart=False)
            assert fresh == want
            inc = _groups(L, h, sel, nlive, focussed=True, restart=first)
            assert inc == want
            nsplit += len(want) > 1
            first = False
            # a few data sets leave: single ones, and now and then a whole slice
            leave = rng.uniform(size=len(sel)) < rng.choice([0.02, 0.1, 0.3])
            if not leave.any():
                leave[rng.randint(len(sel))] = True
            sel = sel[~leave]
    stats = np.zeros(len(core.COUNTERS), dtype=np.int64)
    L.mdns_core_stats(h, stats.ctypes.data)
    s = dict(zip(core.COUNTERS, stats.tolist()))
    assert s["inc_builds"] == 4 and s["inc_updates"] > 10
    assert nsplit > 0, "no selection ever fell into several groups: the test would be vacuous"
    L.mdns_core_destroy(h)
