"""Unit tests of the array-backed bookkeeping against the reference's list-based statement."""
import numpy as np
import pytest

from massivedatans_amd.multi_nested_sampler import _Shelves, find_nsmallest
from massivedatans_amd.clustering.sdml import IdentityMetric, SimpleScaling, TruncatedScaling
from massivedatans_amd import parallel


def test_shelves_match_list_model():
    """Random append / purge / pop sequences: the padded arrays behave like the reference's
    list of per-data-set FIFO lists (multi_nested_sampler.py:117,137-138,482-485,513)."""
    rng = np.random.RandomState(0)
    nd = 37
    sh = _Shelves(nd, cap=2)
    model = [[] for _ in range(nd)]
    pid = 100
    for step in range(400):
        op = rng.randint(3)
        if op == 0:                                   # a new point lands on some shelves
            rows = np.flatnonzero(rng.uniform(size=nd) < 0.3)
            Ls = rng.normal(size=len(rows))
            sh.append(rows, pid, Ls)
            for r, L in zip(rows, Ls):
                model[r].append((pid, L))
            pid += 1
        elif op == 1:                                 # thresholds rise: purge keeps order
            Lmins = rng.normal(size=nd) - 0.5
            sh.purge(Lmins)
            model = [[e for e in shelf if e[1] > Lmins[d]] for d, shelf in enumerate(model)]
        elif all(len(s) > 0 for s in model):          # advance: every data set pops its head
            p, L = sh.pop_heads()
            for d in range(nd):
                want = model[d].pop(0)
                assert (p[d], L[d]) == want
        assert sh.n.tolist() == [len(s) for s in model]
        for d in range(nd):
            assert [(int(sh.p[d, k]), sh.L[d, k]) for k in range(sh.n[d])] == model[d]
            assert np.all(np.isinf(sh.L[d, sh.n[d]:]))
        assert np.array_equal(sh.empty(), np.array([len(s) == 0 for s in model]))
    keep = rng.uniform(size=nd) < 0.5
    sh.select(keep)
    assert sh.n.tolist() == [len(s) for s, k in zip(model, keep) if k]


def test_find_nsmallest():
    rng = np.random.RandomState(1)
    for _ in range(50):
        a, b = rng.normal(size=20), rng.normal(size=rng.randint(1, 6))
        n = len(b)
        assert find_nsmallest(n, a, b) == np.sort(np.concatenate((a, b)))[n]


def test_truncated_scaling_direction():
    """the widest axis gets scale 1, narrower axes get LARGER power-of-two
    divisors (std [0.28, 0.0099, 0.097] -> scale [1, 16, 2])."""
    rng = np.random.RandomState(2)
    X = rng.normal(size=(4000, 3)) * np.array([0.28, 0.0099, 0.097])
    m = TruncatedScaling()
    m.fit(X - X.mean(axis=0))
    assert m.scale.tolist() == [1.0, 16.0, 2.0]
    y = m.transform(X)
    assert np.allclose(m.untransform(y), X)
    s = SimpleScaling()
    s.fit(X)
    assert np.allclose(s.transform(X).std(axis=0), 1.0)
    assert IdentityMetric() == IdentityMetric() and not (IdentityMetric() == m)


def test_shard_helpers():
    b = parallel.shard_bounds(10007, 8)
    assert b[0] == 0 and b[-1] == 10007 and np.all(np.diff(b) >= 1250) and np.all(np.diff(b) <= 1251)
    assert [parallel.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_npz_roundtrip_keeps_reference_dataset_names(tmp_path):
    from massivedatans_amd import gen
    data = gen.horns(7)
    path = str(tmp_path / "data_widths_7.npz")
    gen.save(path, data)
    back = gen.load(path, 5)
    assert {"x", "y", "z"} <= set(back)                      # gensimple_horns.py:61-64
    assert back["y"].shape == (200, 5) and np.array_equal(back["y"], data["y"][:, :5])
    assert np.array_equal(back["x"], data["x"])


class _FakeH5File(dict):
    """Records what would be written through h5py (not installed in this image)."""
    store = {}

    def __init__(self, path, mode):
        super().__init__()
        self.path, self.mode = path, mode
        if mode == 'r':
            self.update({k: v for k, (v, _) in _FakeH5File.store[path].items()})

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def create_dataset(self, name, data=None, **kw):
        _FakeH5File.store.setdefault(self.path, {})[name] = (np.asarray(data), kw)


def test_hdf5_container_uses_reference_names_and_filters(tmp_path, monkeypatch):
    """With h5py present, .hdf5 paths go through it with the reference's dataset names and
    gzip + shuffle (sample.py:202-211, gensimple_horns.py:61-67)."""
    import sys, types
    from massivedatans_amd import gen, sample
    fake = types.ModuleType("h5py")
    fake.File = _FakeH5File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    data = gen.nothing(4)
    path = str(tmp_path / "data_nothing_4.hdf5")
    gen.save(path, data)
    for name, (arr, kw) in _FakeH5File.store[path].items():
        assert kw == dict(compression='gzip', shuffle=True), name
    assert np.array_equal(gen.load(path, 3)["y"], data["y"][:, :3])

    class S:
        ndraws, nevals = 12, 48
    w = [[np.zeros((4, 3)), np.zeros((4, 3)), np.zeros(4), np.zeros(4), np.ones(4, bool)]] * 2
    prefix = str(tmp_path / "out")
    sample.save_results(prefix, dict(logZ=np.zeros(4), logZerr=np.ones(4), weights=w), S, 1.5, 4)
    written = _FakeH5File.store[prefix + ".hdf5"]
    assert set(written) == {"logZ", "logZerr", "u", "x", "L", "w", "mask", "ndraws"}
    assert written["ndraws"][1] == {} and written["u"][0].shape == (2, 4, 3)
    import json
    stats = json.load(open(prefix + ".stats.json"))
    assert stats["ndraws"] == 12 and stats["ndata"] == 4 and stats["niter"] == 2


def test_hdf5_without_h5py_fails_loudly(tmp_path):
    import pytest
    from massivedatans_amd import gen
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed")
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match="h5py"):
        gen.load(str(tmp_path / "x.hdf5"))


def _fake_sampler(lp, npoints, nlive):
    """Just enough of a MultiNestedSampler for the grouping code."""
    from massivedatans_amd.multi_nested_sampler import MultiNestedSampler
    s = MultiNestedSampler.__new__(MultiNestedSampler)
    s.live_pointsp = lp
    s.pointpile = np.zeros((npoints, 3))
    s.nlive_points = nlive
    s.superpoints = set()
    s.point_data_map = None
    s.ndata = lp.shape[1]
    s._lpT = None
    s._walk = None
    s._walk_stale = True
    s._device_groups_wanted = False
    s._dgroups = None
    s._last_selection = None
    s._real_indices = None
    s.real_data_mask_all = np.ones(lp.shape[1], dtype=bool)
    s._refcount = np.bincount(lp.ravel(), minlength=npoints)
    return s


def test_native_grouping_walk_equals_python_walk():
    """csrc/host_groups.c against the Python statement of the reference's walk
    (multi_nested_sampler.py:237-266): same groups in the same order, same point order --
    on id matrices with one big component, many small ones, singletons and full selections."""
    from massivedatans_amd import multi_nested_sampler as mns
    if mns._host_lib() is None:
        import pytest
        pytest.skip("libmdns_host.so not built")
    rng = np.random.RandomState(3)
    for trial in range(40):
        nlive = int(rng.randint(2, 9))
        ndata = int(rng.randint(2, 60))
        nclusters = int(rng.randint(1, 8))
        # data sets draw their live points from the pool of their cluster (+ sometimes a shared one)
        pools = [np.arange(c * 50, c * 50 + rng.randint(nlive, 50)) for c in range(nclusters)]
        lp = np.empty((nlive, ndata), dtype=np.int64)
        for d in range(ndata):
            pool = pools[rng.randint(nclusters)]
            if rng.uniform() < 0.2:
                pool = np.arange(1000 + 10 * d, 1000 + 10 * d + nlive)        # a loner
            lp[:, d] = rng.choice(pool, size=nlive, replace=False)
        npoints = int(lp.max()) + 1
        mask = rng.uniform(size=ndata) < rng.choice([0.3, 0.7, 1.0])
        if mask.sum() < 2:
            mask[:2] = True
        s = _fake_sampler(lp, npoints, nlive)
        native = [(m.copy(), np.asarray(p).copy()) for m, p in s.generate_subsets_nograph(mask, None)]
        lib, mns._host._LIB = mns._host._LIB, False                            # force the Python walk
        try:
            python = [(m.copy(), np.asarray(p).copy()) for m, p in s.generate_subsets_nograph(mask, None)]
        finally:
            mns._host._LIB = lib
        assert len(native) == len(python)
        for (m1, p1), (m2, p2) in zip(native, python):
            assert np.array_equal(m1, m2)
            assert np.array_equal(p1, p2)
        assert np.array_equal(np.sum([m for m, _ in native], axis=0) > 0, mask)


def test_bootstrap_choice_consumes_the_rng_like_the_reference():
    """draw_bootstrap_choice against the reference's spelled-out loop (neighbors.py:170-174):
    same matrix, and the global legacy RNG stream left at the same position."""
    from massivedatans_amd.clustering.neighbors import draw_bootstrap_choice, draw_bootstrap_masks, unpack_bootstrap_masks
    for n in list(range(1, 60)) + [100, 257, 400, 3000]:
        for B in (1, 10, 15):
            np.random.seed(n * 31 + B)
            want = np.zeros((n, B))
            for b in range(B):
                want[np.random.choice(np.arange(n), size=n, replace=True), b] = 1.
            after_want = np.random.uniform()
            np.random.seed(n * 31 + B)
            got = draw_bootstrap_choice(n, B)
            assert np.array_equal(got, want) and np.random.uniform() == after_want
            np.random.seed(n * 31 + B)
            masks = draw_bootstrap_masks(n, B)
            assert np.random.uniform() == after_want
            assert np.array_equal(unpack_bootstrap_masks(masks, B), want)


def test_bootstrap_choice_without_avx512():
    """The same check with the AVX-512 form of the tempering / packing loop switched off (the
    library decides once per process: a child process with MDNS_HOST_NO_AVX512=1), so that both
    forms are compared with numpy wherever the tests run."""
    import os, subprocess, sys
    if os.environ.get("MDNS_HOST_NO_AVX512") == "1":
        pytest.skip("already the scalar form")
    env = dict(os.environ, MDNS_HOST_NO_AVX512="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                          __file__ + "::test_bootstrap_choice_consumes_the_rng_like_the_reference"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_incremental_grouping_walk_on_shrinking_selections():
    """The passes of one iteration ask for the groups of smaller and smaller selections of the
    same id matrix: the incremental native walk (one holder index per iteration, counts kept up
    to date) against the stateless one and against the Python statement, pass by pass --
    including selections that are NOT subsets of the previous one (a new base) and the cases
    without decomposition (few distinct ids; superpoints)."""
    from massivedatans_amd import multi_nested_sampler as mns
    lib = mns._host_lib()
    if lib is None:
        import pytest
        pytest.skip("libmdns_host.so not built")
    rng = np.random.RandomState(8)
    splits = 0
    for trial in range(30):
        nlive = int(rng.randint(2, 9))
        ndata = int(rng.randint(4, 90))
        nclusters = int(rng.randint(1, 6))
        pools = [np.arange(c * 60, c * 60 + rng.randint(nlive, 60)) for c in range(nclusters)]
        lp = np.empty((nlive, ndata), dtype=np.int64)
        for d in range(ndata):
            pool = pools[rng.randint(nclusters)]
            if rng.uniform() < 0.15:
                pool = np.arange(2000 + 10 * d, 2000 + 10 * d + nlive)
            lp[:, d] = rng.choice(pool, size=nlive, replace=False)
        s = _fake_sampler(lp, int(lp.max()) + 1, nlive)
        if trial % 7 == 3:
            s.superpoints = {int(lp[0, 0])}                  # "some points are shared by all"
        mask = np.ones(ndata, dtype=bool)
        for step in range(25):
            if mask.sum() < 2:
                break
            inc = [(m.copy(), np.asarray(p).copy()) for m, p in s._groups_native(lib, mask)]
            ref = [(m.copy(), np.asarray(p).copy()) for m, p in s._groups_native_stateless(lib, mask)]
            held = None
            triv, allp, held = s._trivial_groups(mask, None)
            py = triv if triv is not None else list(s._walk_python(mask, held))
            assert len(inc) == len(ref) == len(py)
            for (m1, p1), (m2, p2), (m3, p3) in zip(inc, ref, py):
                assert np.array_equal(m1, m2) and np.array_equal(p1, p2)
                assert np.array_equal(m1, m3) and np.array_equal(p1, np.asarray(p3))
            splits += len(inc) > 1
            if step % 9 == 8:
                mask = rng.uniform(size=ndata) < 0.8         # not a subset: a new base
            else:
                leave = rng.choice(np.flatnonzero(mask), size=max(1, int(mask.sum() * rng.choice([0.05, 0.3]))), replace=False)
                mask = mask.copy()
                mask[leave] = False
    assert splits > 10


def test_graph_grouping_against_networkx():
    """``generate_subsets_graph`` against an independent implementation: connected components of
    the bipartite (data set, live point) graph from networkx (the cross-check recipe of the
    reference's profile_generate_subsets.py:140-161), put in igraph's documented order --
    components by lowest data-set index, point ids ascending."""
    import pytest
    nx = pytest.importorskip("networkx")
    rng = np.random.RandomState(17)
    multi = 0
    for trial in range(40):
        nlive = int(rng.randint(2, 8))
        ndata = int(rng.randint(3, 50))
        nclusters = int(rng.randint(1, 6))
        pools = [np.arange(c * 40, c * 40 + rng.randint(nlive, 40)) for c in range(nclusters)]
        lp = np.empty((nlive, ndata), dtype=np.int64)
        for d in range(ndata):
            lp[:, d] = rng.choice(pools[rng.randint(nclusters)], size=nlive, replace=False)
        s = _fake_sampler(lp, int(lp.max()) + 1, nlive)
        mask = rng.uniform(size=ndata) < rng.choice([0.5, 1.0])
        if mask.sum() < 2:
            mask[:2] = True
        got = [(np.flatnonzero(m).tolist(), [int(p) for p in pts]) for m, pts in s.generate_subsets_graph(mask, None)]
        G = nx.Graph()
        for d in np.flatnonzero(mask):
            for p in lp[:, d]:
                G.add_edge(("n", int(d)), ("p", int(p)))
        comps = sorted(([sorted(v for k, v in c if k == "n"), sorted(v for k, v in c if k == "p")]
                        for c in nx.connected_components(G)), key=lambda c: c[0][0])
        allp = sorted(set(int(p) for p in lp[:, mask].ravel()))
        if len(comps) == 1 or len(allp) < 2 * nlive:
            want = [(np.flatnonzero(mask).tolist(), allp)]          # multi_nested_sampler.py:283-300,333-335
        else:
            want = [(c[0], c[1]) for c in comps]
        assert got == want
        multi += len(want) > 1
    assert multi > 5


def _split_with_superpoint(nlive=6):
    """Two blocks of data sets with DISJOINT live points and enough distinct ids (>= 2 nlive) --
    plus a point recorded as a superpoint although it is live nowhere: it landed on every shelf
    (multi_nested_sampler.py:486-488) and is not live anywhere yet."""
    half = nlive // 2
    cols = [np.arange(nlive), np.arange(nlive) + half]                          # data sets 0, 1 overlap
    cols += [100 + np.arange(nlive) + k * half for k in range(3)]               # data sets 2, 3, 4 chain
    lp = np.array(cols, dtype=np.int64).T
    return lp, 200


def _graph_groups(s, mask):
    return [(np.flatnonzero(np.asarray(g[0])).tolist(), [int(p) for p in g[1]]) for g in s.generate_subsets_graph(mask, None)]


def test_superpoint_shortcut_holds_while_the_graph_is_split():
    """multi_nested_sampler.py:284-297: with superpoints known the reference returns ONE group
    (all selected data sets, numpy.unique of their ids) without looking at the graph -- also when
    the live-id graph is split, which can happen because a point enters `superpoints` on landing
    on all shelves, before it is live.  The native host path and the Python path must both do so
    (the device path: tests/test_groups.py)."""
    from massivedatans_amd import multi_nested_sampler as mns
    lp, npoints = _split_with_superpoint()
    mask = np.ones(lp.shape[1], dtype=bool)
    allp = sorted(set(int(p) for p in lp.ravel()))
    for planted in (False, True):
        want_one = [(list(range(lp.shape[1])), allp)]
        s = _fake_sampler(lp, npoints, lp.shape[0])
        if planted:
            s.superpoints = {150}                       # on every shelf, live nowhere
        native = _graph_groups(s, mask)
        lib, mns._host._LIB = mns._host._LIB, False     # the Python statement
        try:
            s2 = _fake_sampler(lp, npoints, lp.shape[0])
            if planted:
                s2.superpoints = {150}
            python = _graph_groups(s2, mask)
        finally:
            mns._host._LIB = lib
        assert native == python
        if planted:
            assert native == want_one
        else:
            assert len(native) == 2 and native[0][0] == [0, 1] and native[1][0] == [2, 3, 4]


def _muse_case(seed=4, nd=23, nx=61):
    rng = np.random.RandomState(seed)
    x = np.linspace(4750, 9350, nx)
    y = rng.normal(1.0, 0.1, size=(nx, nd))
    v = rng.uniform(0.5, 2.0, size=(nx, nd)) * 1e-2
    ypred = 1.0 + 0.3 * np.exp(-0.5 * ((x - 6000.) / 300.) ** 2)
    mask = rng.uniform(size=nd) < 0.6
    mask[0] = True
    return x, np.ascontiguousarray(y), np.ascontiguousarray(v), ypred, mask


def _check_muse_jitter(spectra, oracle, y, v, ypred, mask, rtol):
    """musefuse.py:534-535: ``Lout[data_mask] + numpy.random.normal(0, 1e-5, size=data_mask.sum())``
    -- the values AND the position of the global RNG stream afterwards."""
    Lout = np.zeros(y.shape[1])
    oracle.muse_like(y, v, ypred, mask, Lout=Lout)
    np.random.seed(77)
    want = Lout[mask] + np.random.normal(0, 1e-5, size=mask.sum())
    probe_want = np.random.uniform()
    np.random.seed(77)
    got = spectra.multi_loglikelihood(ypred, mask, jitter=np.random.normal)
    assert np.random.uniform() == probe_want, "the jitter must consume exactly mask.sum() normal deviates"
    assert got.shape == want.shape
    assert np.max(np.abs(got - want) / np.abs(want)) < rtol
    # without jitter: the bare likelihoods, no RNG consumed
    np.random.seed(78)
    bare = spectra.multi_loglikelihood(ypred, mask)
    np.random.seed(78)
    assert np.max(np.abs(bare - Lout[mask]) / np.abs(Lout[mask])) < rtol
    # "no stars" (musefuse.py:527-529): constant -1e100, and NO random numbers drawn
    np.random.seed(79)
    probe = np.random.uniform()
    np.random.seed(79)
    flat = spectra.multi_loglikelihood(np.zeros_like(ypred), mask, jitter=np.random.normal)
    assert np.array_equal(flat, np.ones(mask.sum()) * -1e100) and np.random.uniform() == probe


def test_muse_multi_loglikelihood_jitter_host_logic(oracle):
    """The K2 caller (like.MuseSpectra.multi_loglikelihood, musefuse.py:520-535) with the oracle
    standing in for the kernel: tie-breaking jitter, its RNG consumption, the no-stars case."""
    from massivedatans_amd.like import MuseSpectra
    x, y, v, ypred, mask = _muse_case()

    class OracleMuse(MuseSpectra):
        def __init__(self):                   # no device: only the host logic is under test
            self.ndata, self.nx = y.shape[1], y.shape[0]

        def loglike_batch(self, templates, data_mask=None):
            out = np.zeros(self.ndata)
            oracle.muse_like(y, v, np.ascontiguousarray(np.atleast_2d(templates)[0]), data_mask, Lout=out)
            return out[data_mask][None, :]

        def close(self):
            pass

    _check_muse_jitter(OracleMuse(), oracle, y, v, ypred, mask, rtol=1e-15)


def test_hdf5_roundtrip_with_real_h5py(tmp_path):
    """Input and output files through the real h5py where it is installed (it is not in the
    build image: the test skips there): dataset names of gensimple_horns.py:61-67 and
    sample.py:202-211, gzip + shuffle filters, values back bit for bit."""
    import pytest
    h5py = pytest.importorskip("h5py")
    from massivedatans_amd import gen, sample
    data = gen.horns(6)
    path = str(tmp_path / "data_widths_6.hdf5")
    gen.save(path, data)
    with h5py.File(path, "r") as f:
        assert {"x", "y", "z"} <= set(f.keys())
        assert f["y"].compression == "gzip" and f["y"].shuffle
        assert np.array_equal(f["y"][()], data["y"])
    assert np.array_equal(gen.load(path, 4)["y"], data["y"][:, :4])

    class S:
        ndraws, nevals = 12, 48
    w = [[np.zeros((6, 3)), np.ones((6, 3)), np.arange(6.), np.zeros(6), np.ones(6, bool)]] * 3
    prefix = str(tmp_path / "out")
    sample.save_results(prefix, dict(logZ=np.arange(6.), logZerr=np.ones(6), weights=w), S, 1.5, 6)
    with h5py.File(prefix + ".hdf5", "r") as f:
        assert set(f.keys()) == {"logZ", "logZerr", "u", "x", "L", "w", "mask", "ndraws"}
        assert f["u"].shape == (3, 6, 3) and f["L"].compression == "gzip" and f["L"].shuffle
        assert int(f["ndraws"][()]) == 12 and np.array_equal(f["logZ"][()], np.arange(6.))


@pytest.mark.gpu
def test_hdf5_roundtrip_with_real_h5py_on_the_gpu_box(tmp_path):
    """The same round trip in the ``-m gpu`` tier, so that a recorded run on the GPU box says whether
    h5py was there: it SKIPS with that reason when it is not (SURVEY row f4 stays partial until one
    recorded run executes it)."""
    try:
        import h5py  # noqa: F401
    except ImportError:
        pytest.skip("h5py is not installed on this box: the real-file round trip (row f4) did not run")
    test_hdf5_roundtrip_with_real_h5py(tmp_path)

