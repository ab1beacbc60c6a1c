"""Grouping of data sets on the device (include/mdns.h Part 4, csrc/mdns_groups.hip) against
independent CPU statements: scipy's connected components of the same bipartite graph and
numpy.unique of the selected columns -- the two things the reference's generate_subsets_graph
gets from igraph and numpy (multi_nested_sampler.py:268-355).  Integer results: exact."""
import hashlib
import json
import os

import numpy as np
import pytest

from massivedatans_amd import gen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_groups(lp, rows):
    """[(member data sets, distinct ids)], components in order of their lowest data set, ids
    ascending: scipy.sparse.csgraph on the bipartite graph of the selected columns."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    rows = np.asarray(rows)
    sub = lp[:, rows]
    ids, inv = np.unique(sub, return_inverse=True)
    inv = inv.reshape(sub.shape)
    nd, npnt = len(rows), len(ids)
    r = np.repeat(np.arange(nd)[None, :], sub.shape[0], axis=0).ravel()
    c = nd + inv.ravel()
    g = coo_matrix((np.ones(len(r)), (r, c)), shape=(nd + npnt, nd + npnt))
    _, lab = connected_components(g, directed=False)
    out = []
    seen = []
    for k in lab[:nd]:
        if k not in seen:
            seen.append(k)                       # first appearance = lowest data set of the component
    for k in seen:
        out.append((rows[lab[:nd] == k], ids[lab[nd:] == k]))
    return out


def clustered_ids(rng, nlive, ndata, nclusters, pool):
    pools = [np.arange(c * pool, c * pool + rng.randint(nlive, pool)) for c in range(nclusters)]
    lp = np.empty((nlive, ndata), dtype=np.int64)
    for d in range(ndata):
        lp[:, d] = rng.choice(pools[rng.randint(nclusters)], size=nlive, replace=False)
    return lp


def same(got, want):
    assert len(got) == len(want)
    for (m1, p1), (m2, p2) in zip(got, want):
        assert np.array_equal(m1, m2) and np.array_equal(p1, p2)


def test_components_against_scipy_small_and_ragged():
    from massivedatans_amd.grouping import DeviceGroups
    rng = np.random.RandomState(5)
    multi = 0
    for trial in range(30):
        nlive = int(rng.randint(1, 9))
        ndata = int(rng.randint(1, 70))
        lp = clustered_ids(rng, nlive, ndata, int(rng.randint(1, 6)), 40)
        npoints = int(lp.max()) + 1 + int(rng.randint(0, 100))
        dg = DeviceGroups(lp)
        assert np.array_equal(dg.ids(), lp)
        for sel in range(3):
            if sel == 0:
                rows = None
                want = cpu_groups(lp, np.arange(ndata))
            else:
                rows = np.flatnonzero(rng.uniform(size=ndata) < rng.choice([0.2, 0.7]))
                if len(rows) == 0:
                    rows = np.array([ndata - 1])
                want = cpu_groups(lp, rows)
            got = dg.groups(rows, npoints)
            same(got, want)
            multi += len(want) > 1
            # the labels of the listed ids (one call's worth) are those of the full label array
            ncomp, ids = dg.components(rows, npoints)
            labels_a, of_ids = dg.labels_of_ids(len(ids))
            ncomp2, ids2 = dg.components(rows, npoints)
            labels_b, of_points = dg.labels()
            assert ncomp == ncomp2 == len(want) and np.array_equal(ids, ids2)
            assert np.array_equal(labels_a, labels_b) and np.array_equal(of_ids, of_points[ids])
        dg.close()
    assert multi > 10


@pytest.mark.parametrize("ndata,nlive,nclusters", [(5000, 100, 1), (10000, 100, 7), (3001, 37, 300), (100000, 20, 3)])
def test_components_against_scipy_full_size(ndata, nlive, nclusters):
    """Sizes of the BASELINE configurations (10 000 x 100 is configs[1]'s id matrix, 100 000 data
    sets configs[3]'s): one big component, a few, hundreds; all data sets and sparse selections;
    twice in a row (the state of the previous call must not leak)."""
    from massivedatans_amd.grouping import DeviceGroups
    rng = np.random.RandomState(ndata + nclusters)
    lp = clustered_ids(rng, nlive, ndata, nclusters, max(4 * nlive, 150))
    npoints = int(lp.max()) + 1
    dg = DeviceGroups(lp)
    for rows in (None, np.flatnonzero(rng.uniform(size=ndata) < 0.3), np.flatnonzero(rng.uniform(size=ndata) < 0.002), None):
        if rows is not None and len(rows) == 0:
            continue
        want = cpu_groups(lp, np.arange(ndata) if rows is None else rows)
        same(dg.groups(rows, npoints), want)
        ncomp, ids = dg.components(rows, npoints)
        assert ncomp == len(want)
        assert np.array_equal(ids, np.unique(lp if rows is None else lp[:, rows]))
    # the same set as a bit map
    bits = dg.touched(npoints)
    held = np.flatnonzero(np.unpackbits(bits.view(np.uint8), bitorder="little")[:npoints])
    assert np.array_equal(held, np.unique(lp))
    dg.close()


@pytest.mark.parametrize("n", [10, 1000, 20000])
def test_long_chains_converge(n):
    """The worst case for label propagation: data set d holds ids d and d + 1, a path whose diameter
    is the number of data sets -- in order and with the data sets shuffled along the path.  One
    component, every id; the rounds stay in the dozens (pointer chase + hooking), and a call is not
    limited to one batch of rounds."""
    from massivedatans_amd.grouping import DeviceGroups
    rng = np.random.RandomState(n)
    lp = np.vstack([np.arange(n), np.arange(n) + 1])
    for mat in (lp, lp[:, rng.permutation(n)]):
        dg = DeviceGroups(mat)
        ncomp, ids = dg.components(None, n + 1)
        assert ncomp == 1 and np.array_equal(ids, np.arange(n + 1))
        assert dg.mean_rounds() < 64
        # and cut in two: the data sets holding id n // 2 left out
        keep = np.flatnonzero(~(mat == n // 2).any(axis=0))
        if len(keep) >= 2:
            same(dg.groups(keep, n + 1), cpu_groups(mat, keep))
        dg.close()


def test_replacements_follow_the_host_matrix():
    """mdns_groups_replace (the end of an iteration: one id per running data set changes) keeps
    the device matrix equal to the host's, and the components follow -- a bridge point joins two
    components, its removal splits them again."""
    from massivedatans_amd.grouping import DeviceGroups
    rng = np.random.RandomState(9)
    nlive, ndata = 12, 400
    lp = clustered_ids(rng, nlive, ndata, 2, 60)
    npoints = 400
    dg = DeviceGroups(lp)
    assert len(dg.groups(None, npoints)) == 2
    for it in range(25):
        rows = np.flatnonzero(rng.uniform(size=ndata) < 0.5)
        slots = rng.randint(0, nlive, size=len(rows))
        new = 200 + it * 2 + rng.randint(0, 2, size=len(rows))      # ids shared across both components
        lp[slots, rows] = new
        dg.replace(rows, slots, new)
        assert np.array_equal(dg.ids(), lp)
        same(dg.groups(None, npoints), cpu_groups(lp, np.arange(ndata)))
        sel = np.flatnonzero(rng.uniform(size=ndata) < 0.1)
        if len(sel):
            same(dg.groups(sel, npoints), cpu_groups(lp, sel))
    dg.close()


def test_superpoint_shortcut_on_the_device_path():
    """`superpoints` non-empty while the live-id graph is split (a point that landed
    on every shelf and is live nowhere yet): the reference returns one joint group
    (multi_nested_sampler.py:284-297); the device branch of generate_subsets_graph must too."""
    from test_sampler_units import _fake_sampler, _graph_groups, _split_with_superpoint
    lp, npoints = _split_with_superpoint()
    mask = np.ones(lp.shape[1], dtype=bool)
    allp = sorted(set(int(p) for p in lp.ravel()))
    for planted in (False, True):
        host = _fake_sampler(lp, npoints, lp.shape[0])
        dev = _fake_sampler(lp, npoints, lp.shape[0])
        dev._device_groups_wanted = True
        if planted:
            host.superpoints = {150}
            dev.superpoints = {150}
        got = _graph_groups(dev, mask)
        assert dev._dgroups is not None and dev._dgroups.ncalls == 1
        assert got == _graph_groups(host, mask)
        assert (got == [(list(range(lp.shape[1])), allp)]) == planted
        dev._dgroups.close()


def test_bad_arguments_are_refused():
    from massivedatans_amd import _lib
    from massivedatans_amd.grouping import DeviceGroups
    lp = np.arange(12).reshape(3, 4)
    dg = DeviceGroups(lp)
    with pytest.raises(_lib.MdnsError):
        dg.components(np.array([2, 1]), 12)              # not ascending
    with pytest.raises(_lib.MdnsError):
        dg.components(np.array([0, 4]), 12)              # outside
    with pytest.raises(_lib.MdnsError):
        dg.components(None, 5)                           # ids beyond npoints
    ncomp, ids = dg.components(None, 12)                 # and the handle still works
    assert ncomp == 4 and np.array_equal(ids, np.arange(12))
    dg.close()


@pytest.mark.parametrize("key", ["horns-graph_100_40_400", "nothing-graph_100_40_400", "horns-graph_300_40_300"])
def test_graph_variant_on_the_gpu_matches_the_cpu_path(key, monkeypatch):
    """USE_GRAPH=1 (the reference's default grouping) end to end: the GPU run -- components and
    distinct ids from the device (csrc/mdns_groups.hip), draws decided on the device -- against
    the host orchestration on the CPU oracle backends, whose graph grouping is native host code
    (tests/golden/bookkeeping_c2.json, oracle/make_bookkeeping_hash.py): same draws, the pile of
    accepted points byte for byte, evidences within 1e-9."""
    from massivedatans_amd import sample
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "bookkeeping_c2.json")))
    if key not in table:
        pytest.skip("fixture not generated (oracle/make_bookkeeping_hash.py --case %s)" % key.replace("_", ":"))
    want = table[key]
    kind, ndata, nlive, cap = key.split("_")
    data = (gen.horns if kind.startswith("horns") else gen.nothing)(int(ndata))
    # (the sampler core groups small selections on the host: here every selection goes to the device)
    monkeypatch.setenv("MDNS_CORE_HOST_EDGES", "0")
    with np.errstate(all="ignore"):
        results, sampler, _, _ = sample.run(data["x"], data["y"], nlive_points=int(nlive), max_samples=int(cap), use_graph=True)
    assert sampler._dgroups is not None and sampler._dgroups.ncalls > 0
    assert sampler.ndraws == want["ndraws"]
    assert len(sampler.pointpile) == want["npoints"]
    got = hashlib.sha256(np.ascontiguousarray(sampler.pointpile, dtype=np.float64).tobytes()).hexdigest()
    assert got == want["pointpile_sha256"]
    assert np.max(np.abs(results["logZ"][:5] - np.array(want["logZ_first5"]))) < 1e-9


@pytest.mark.parametrize("kind,ndata,nlive,cap", [("horns", 10000, 100, 520), ("nothing", 3000, 60, 0)])
def test_device_grouping_equals_host_grouping_end_to_end(kind, ndata, nlive, cap, monkeypatch):
    """The same GPU analysis with USE_GRAPH=1 twice: components and ids from the device
    (csrc/mdns_groups.hip) and from the native host code (csrc/host_groups.c + sort,
    MDNS_DEVICE_GROUPS=0) -- two independent implementations of generate_subsets_graph's
    partition: every draw, every accepted point and every evidence must coincide."""
    from massivedatans_amd import sample
    data = (gen.horns if kind == "horns" else gen.nothing)(ndata)
    runs = []
    monkeypatch.setenv("MDNS_CORE_HOST_EDGES", "0")     # with a device grouping, every selection goes there
    for device in ("1", "0"):
        monkeypatch.setenv("MDNS_DEVICE_GROUPS", device)
        with np.errstate(all="ignore"):
            results, sampler, _, _ = sample.run(data["x"], data["y"], nlive_points=nlive, max_samples=cap, use_graph=True)
        assert (sampler._dgroups is not None) == (device == "1")
        runs.append((int(sampler.ndraws), np.ascontiguousarray(sampler.pointpile).tobytes(), results["logZ"].copy(),
                     sampler._dgroups.ncalls if sampler._dgroups is not None else 0))
        if sampler.joint is not None:
            sampler.joint.close()
    assert runs[0][3] > 100                               # the device was asked
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    assert np.array_equal(runs[0][2], runs[1][2])
