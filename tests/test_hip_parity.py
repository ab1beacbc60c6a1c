"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the
committed golden vectors (made from the reference's own C).

Bars: geometry (K3-K6) bit-exact; likelihoods (K1, K2) within 1e-6 relative as BASELINE.json
states -- the tests assert the much tighter 1e-12 the kernels actually reach.
"""
import ctypes as C
import os

import numpy as np
import pytest

from massivedatans_amd import _lib, gen

pytestmark = pytest.mark.gpu

RTOL_L = 1e-12      # asserted; the contract in BASELINE.json is 1e-6 relative


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) if a.size else 0.0


@pytest.fixture(scope="module")
def nb():
    from massivedatans_amd.clustering import neighbors
    return neighbors


# ---------------------------------------------------------------- K1 ----------------------
def test_k1_dropin_matches_golden(hip, golden):
    for name in ("horns", "nothing"):
        x, y = golden["k1_%s_x" % name], golden["k1_%s_y" % name]
        nx, nd = y.shape
        for mi, m in enumerate(golden["k1_%s_masks" % name]):
            m = np.ascontiguousarray(m)
            want = golden["k1_%s_out%d" % (name, mi)]
            for p, w in zip(golden["k1_%s_params" % name], want):
                out = np.zeros(int(m.sum()))
                rc = hip.mdns_gauss_like(_lib.ptr(x), _lib.ptr(y), nd, nx, p[0], p[1], 10 ** p[2], 0.01,
                                         _lib.ptr(m), _lib.ptr(out))
                assert rc == 0, hip.mdns_last_error()
                assert out.shape == w.shape
                assert rel_err(out, w) < RTOL_L
    # += semantics of clike.c:72
    pre = golden["k1_accum_pre"].copy()
    x, y = golden["k1_horns_x"], golden["k1_horns_y"]
    m = np.ones(y.shape[1], dtype=np.bool_)
    assert hip.mdns_gauss_like(_lib.ptr(x), _lib.ptr(y), y.shape[1], y.shape[0], 0.3, 640., 4., 0.01,
                               _lib.ptr(m), _lib.ptr(pre)) == 0
    assert rel_err(pre, golden["k1_accum_out"]) < RTOL_L


def test_k1_dropin_shim_library(golden):
    """Through the drop-in clike.so exactly as sample.py:84-106 binds it."""
    lib = C.CDLL(os.path.join(_lib.DROPIN_DIR, "clike.so"))
    nd_ = np.ctypeslib.ndpointer
    lib.like.argtypes = [nd_(dtype=np.float64, ndim=1, flags='C_CONTIGUOUS'),
                         nd_(dtype=np.float64, ndim=2, flags='C_CONTIGUOUS'), C.c_int, C.c_int,
                         C.c_double, C.c_double, C.c_double, C.c_double,
                         nd_(dtype=np.bool_, ndim=1, flags='C_CONTIGUOUS'),
                         nd_(dtype=np.float64, ndim=1, flags='C_CONTIGUOUS')]
    x, y = golden["k1_horns_x"], golden["k1_horns_y"]
    m = np.ascontiguousarray(golden["k1_horns_masks"][1])
    p = golden["k1_horns_params"][0]
    Lout = np.zeros(m.sum())
    assert lib.like(x, y, y.shape[1], y.shape[0], p[0], p[1], 10 ** p[2], 0.01, m, Lout) == 0
    assert rel_err(Lout, golden["k1_horns_out1"][0]) < RTOL_L


@pytest.mark.parametrize("nd,nx", [(1, 1), (3, 7), (130, 128), (257, 200), (64, 255), (40, 300),
                                   (33, 512), (9, 777)])
def test_k1_batch_vs_oracle_ragged(oracle, nd, nx):
    from massivedatans_amd.like import GaussLineSpectra
    rng = np.random.RandomState(nd * 1000 + nx)
    x = np.sort(rng.uniform(400, 800, nx))
    y = np.ascontiguousarray(rng.normal(0, 0.05, size=(nx, nd)))
    sp = GaussLineSpectra(x, y, noise_level=0.01)
    for B in (1, 3, 4, 9):
        params = np.column_stack([rng.uniform(0.01, 1, B), rng.uniform(400, 800, B),
                                  10 ** rng.uniform(0, 2, B)])
        for mask in (np.ones(nd, bool), rng.uniform(size=nd) < 0.5, np.zeros(nd, bool)):
            got = sp.loglike_batch(params, mask)
            assert got.shape == (B, mask.sum())
            for b in range(B):
                want = -0.5 * oracle.gauss_like(x, y, params[b, 0], params[b, 1], params[b, 2], 0.01, mask)
                assert rel_err(got[b], want) < RTOL_L
    # reference call signature (sample.py:101-108): params = (A, mu, log10 sig)
    L = sp.multi_loglikelihood(np.array([0.5, 600., 1.0]), np.ones(nd, bool))
    want = -0.5 * oracle.gauss_like(x, y, 0.5, 600., 10.0, 0.01, np.ones(nd, bool))
    assert rel_err(L, want) < RTOL_L
    sp.close()


def test_k1_dataset_major_layout_and_row_ids(oracle):
    from massivedatans_amd.like import GaussLineSpectra
    d = gen.horns(300)
    sp = GaussLineSpectra(d["x"], np.ascontiguousarray(d["y"].T), layout="dataset_major")
    rows = np.array([7, 3, 299, 0, 3], dtype=np.int32)       # any order, repeats allowed
    got = sp.loglike_batch(np.array([[0.2, 650., 5.0]]), rows)[0]
    full = -0.5 * oracle.gauss_like(d["x"], d["y"], 0.2, 650., 5.0, 0.01, np.ones(300, bool))
    assert rel_err(got, full[rows]) < RTOL_L


def test_k1_full_size_properties():
    """Config C2 size (10 000 x 200): size-independent properties instead of the oracle."""
    from massivedatans_amd.like import GaussLineSpectra
    d = gen.horns(10000)
    sp = GaussLineSpectra(d["x"], d["y"])
    rng = np.random.RandomState(5)
    params = np.column_stack([rng.uniform(0.01, 1, 32), rng.uniform(400, 800, 32), 10 ** rng.uniform(0, 2, 32)])
    full = sp.loglike_batch(params)
    assert full.shape == (32, 10000) and np.all(np.isfinite(full)) and np.all(full <= 0)
    # (1) batching is transparent: any split into batches of 12+ candidates runs the same
    # per-spectrum summation and is bitwise identical; small batches take the row kernel
    # (tree sum) and agree to rounding
    assert np.array_equal(sp.loglike_batch(params[:16]), full[:16])
    assert np.array_equal(sp.loglike_batch(params[12:32]), full[12:32])
    assert rel_err(sp.loglike_batch(params[3:9]), full[3:9]) < 1e-13
    for b in (0, 7, 31):
        assert rel_err(sp.loglike_batch(params[b:b + 1])[0], full[b]) < 1e-13
    # (2) masking is a gather of the full result: bitwise for dense selections (same kernel,
    # same per-spectrum summation order), to rounding for sparse ones (row kernel, tree sum)
    m = rng.uniform(size=10000) < 0.5
    assert np.array_equal(sp.loglike_batch(params, m), full[:, m])
    m = rng.uniform(size=10000) < 0.01
    assert np.array_equal(sp.loglike_batch(params, m), full[:, m])          # 32 candidates: lane kernel + gather
    assert rel_err(sp.loglike_batch(params[:20], m), full[:20, m]) < 1e-13  # sparse, < 32: row kernel
    # (3) A -> 0 reproduces the closed-form null evidence of plotevidences.py:17
    null = sp.loglike_batch(np.array([[0.0, 600., 5.0]]))[0]
    want = (-0.5 * (d["y"] / 0.01) ** 2).sum(axis=0)
    assert rel_err(null, want) < 1e-12
    # (4) a spectrum scored against its own noiseless line: chi^2 is that of pure noise
    i = 123
    L = sp.loglike_batch(np.array([[d["height_narrow"][i], d["mean_narrow"][i], 5.0]]))[0][i]
    assert -0.5 * 200 * 2.0 < L < -0.5 * 200 * 0.5


# ---------------------------------------------------------------- K2 ----------------------
def test_k2_dropin_matches_golden(hip, golden):
    yy, vv = golden["k2_y"], golden["k2_v"]
    nx, nd = yy.shape
    for mi, m in enumerate(golden["k2_masks"]):
        m = np.ascontiguousarray(m)
        for yp, w in zip(golden["k2_ypred"], golden["k2_out%d" % mi]):
            L = np.full(nd, 12345.0)
            yp = np.ascontiguousarray(yp)
            assert hip.mdns_muse_like(_lib.ptr(yy), _lib.ptr(vv), _lib.ptr(yp), _lib.ptr(m), nd, nx,
                                      _lib.ptr(L)) == 0, hip.mdns_last_error()
            assert np.all(L[~m] == 12345.0)                   # unmasked entries untouched
            assert rel_err(L[m], w[m]) < 1e-11


@pytest.mark.parametrize("nd,nx", [(5, 3), (17, 96), (9, 513), (6, 1500), (4, 4096), (3, 5000)])
def test_k2_batch_vs_oracle(oracle, nd, nx):
    from massivedatans_amd.like import MuseSpectra
    cube = gen.muse_like(nd, nx=nx)
    sp = MuseSpectra(cube["x"], cube["y"], cube["v"])
    rng = np.random.RandomState(nx)
    pars = np.column_stack([rng.uniform(-0.5, 0.5, 3), rng.uniform(0, 0.02, 3), rng.uniform(-0.1, 0.2, 3),
                            rng.uniform(0.5, 1.5, 3), rng.uniform(0.5, 1.5, 3)])
    ypred = np.array([gen.muse_template(cube["x"], p) for p in pars])
    for mask in (np.ones(nd, bool), np.arange(nd) % 2 == 0):
        got = sp.loglike_batch(ypred, mask)
        got_dev = sp.loglike_batch_lines(pars, mask)          # template evaluated on the device
        for b in range(3):
            want = oracle.muse_like(cube["y"], cube["v"], np.ascontiguousarray(ypred[b]), mask)[mask]
            assert rel_err(got[b], want) < 1e-11
            assert rel_err(got_dev[b], want) < 1e-9
    sp.close()


# ---------------------------------------------------------------- K3-K6 -------------------
def test_k3_k4_golden_bit_exact(hip, nb, golden):
    for ndim in (3, 5):
        t = "k3_d%d" % ndim
        mem, cand, r = golden[t + "_members"], golden[t + "_cands"], float(golden[t + "_r"])
        for cm in (0, 1, 3):
            out = np.zeros(len(cand))
            assert hip.mdns_count_within_distance_of(_lib.ptr(mem), len(mem), ndim, r, _lib.ptr(cand),
                                                     len(cand), _lib.ptr(out), cm) == 0
            assert np.array_equal(out, golden[t + "_count%d" % cm])
        pre = golden[t + "_pre"].copy()
        assert hip.mdns_count_within_distance_of(_lib.ptr(mem), len(mem), ndim, r, _lib.ptr(cand),
                                                 len(cand), _lib.ptr(pre), 2) == 0
        assert np.array_equal(pre, golden[t + "_count2_pre"])
        got = np.array([nb.is_within_distance_of(mem, r, c) for c in cand[:40]])
        assert np.array_equal(got, golden[t + "_any"])
        assert np.array_equal(nb.count_within_distance_of(mem, r, cand), golden[t + "_count0"].astype(int))
        assert np.array_equal(nb.any_within_distance_of(mem, r, cand), golden[t + "_count1"] > 0)
    # strict comparison at distance exactly r
    out = np.zeros(3)
    assert hip.mdns_count_within_distance_of(_lib.ptr(golden["k3_edge_members"]), 1, 3,
                                             float(golden["k3_edge_r"]), _lib.ptr(golden["k3_edge_cands"]),
                                             3, _lib.ptr(out), 0) == 0
    assert np.array_equal(out, golden["k3_edge_count0"])


def test_k5_k6_golden_bit_exact(nb, golden):
    for ndim in (3, 5, 2):
        t = "k6_d%d" % ndim
        pts = golden[t + "_pts"]
        assert nb.most_distant_nearest_neighbor(pts) == float(golden[t + "_nn"])
        for chosen, r in zip(golden[t + "_chosen"], golden[t + "_radius"]):
            assert nb.bootstrapped_maxdistance_chosen(pts, chosen) == r
    assert nb.bootstrapped_maxdistance_chosen(golden["k6_quirk_pts"], golden["k6_quirk_chosen"]) == \
        float(golden["k6_quirk_radius"])


def test_k6_rng_call_order(nb, oracle):
    """bootstrapped_maxdistance draws its chosen matrix from the global legacy stream exactly as
    clustering/neighbors.py:170-174 does."""
    rng = np.random.RandomState(3)
    pts = rng.uniform(size=(150, 3))
    np.random.seed(11)
    got = nb.find_rdistance(pts, nbootstraps=10)
    after = np.random.uniform()
    np.random.seed(11)
    chosen = np.zeros((150, 10))
    for b in range(10):
        chosen[np.random.choice(np.arange(150), size=150, replace=True), b] = 1.
    assert got == oracle.bootstrapped_maxdistance(pts, chosen)
    assert after == np.random.uniform()


@pytest.mark.parametrize("ndim", [1, 2, 3, 4, 5, 8, 11])
def test_geometry_vs_oracle_sweep(nb, oracle, ndim):
    rng = np.random.RandomState(100 + ndim)
    for K, M in ((1, 5), (2, 1), (63, 300), (700, 1000), (1500, 257)):
        pts = rng.uniform(size=(K, ndim))
        cand = rng.uniform(-0.2, 1.2, size=(M, ndim))
        cand[: min(M, K, 3)] = pts[: min(M, K, 3)]
        r = 0.4 * ndim ** 0.5 * rng.uniform(0.2, 1.0)
        assert np.array_equal(nb.count_within_distance_of(pts, r, cand),
                              oracle.count_within_distance_of(pts, r, cand).astype(int))
        assert np.array_equal(nb.any_within_distance_of(pts, r, cand),
                              oracle.count_within_distance_of(pts, r, cand, countmax=1) > 0)
        assert nb.most_distant_nearest_neighbor(pts) == oracle.most_distant_nearest_neighbor(pts)
        for nboot in (1, 10, 19):
            chosen = np.zeros((K, nboot))
            for b in range(nboot):
                chosen[rng.choice(np.arange(K), size=K, replace=True), b] = 1.
            assert nb.bootstrapped_maxdistance_chosen(pts, chosen) == oracle.bootstrapped_maxdistance(pts, chosen)


def test_geometry_full_size_properties(nb):
    """K = 10 000 pool points (config C2 late-run size): properties that need no oracle."""
    rng = np.random.RandomState(9)
    pts = rng.uniform(size=(10000, 3))
    cand = rng.uniform(size=(10000, 3))
    r = 0.05
    counts = nb.count_within_distance_of(pts, r, cand)
    # counting is additive over a split of the members
    a = nb.count_within_distance_of(pts[:3777], r, cand)
    b = nb.count_within_distance_of(pts[3777:], r, cand)
    assert np.array_equal(counts, a + b)
    assert np.array_equal(nb.any_within_distance_of(pts, r, cand), counts > 0)
    # members are within any positive radius of themselves; radius grows monotonically the count
    assert np.all(nb.count_within_distance_of(pts, r, pts[:500]) >= 1)
    assert np.all(nb.count_within_distance_of(pts, 2 * r, cand) >= counts)
    # bootstrap radius: with everything chosen nothing is left out -> 0; K5 >= any K6 round of
    # the full pool is not implied, but K6 is invariant under permuting the rounds
    chosen = np.zeros((10000, 10))
    for k in range(10):
        chosen[rng.choice(np.arange(10000), size=10000, replace=True), k] = 1.
    r6 = nb.bootstrapped_maxdistance_chosen(pts, chosen)
    assert r6 == nb.bootstrapped_maxdistance_chosen(pts, np.ascontiguousarray(chosen[:, ::-1]))
    assert nb.bootstrapped_maxdistance_chosen(pts, np.ones((10000, 3))) == 0.0
    assert 0 < r6 < 1


# ---------------------------------------------------------------- end to end ---------------
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", ["nothing4", "horns3", "horns12", "horns100", "nothing4_graph", "horns12_graph", "horns100_graph"])
def test_end_to_end_against_reference_trace(case, fused, monkeypatch):
    """The whole analysis on the GPU (HIP likelihood + HIP geometry + host orchestration) against
    the trace recorded from the reference's Python + C.  Geometry is bit-exact and likelihoods
    agree to ~1e-15, so the integer bookkeeping is expected to coincide (a last-bit tie in an
    accept test could fork a run; none occurs in these cases): per-iteration draw counts, the
    accepted points and the ids of every data set's live points at its termination EXACTLY, the
    position of the global RNG stream exactly, per-iteration dead-point likelihoods and the
    live-point likelihoods within 1e-12, evidences within the 1e-6 relative bar of BASELINE.json
    (and 1e-9 absolute).  ``fused``: live-point likelihoods, shelves, thresholds, accept test and
    shelf fill on the device (mdns_joint_*).  ``*_graph``: the reference's default grouping
    (USE_GRAPH=1) -- with ``fused`` its components come from the device (csrc/mdns_groups.hip)."""
    from massivedatans_amd import multi_nested_sampler, sample
    from massivedatans_amd.multi_nested_integrator import multi_nested_integrator
    from tracing import Recorder, check_bookkeeping, check_floats, load_trace
    # (every iteration also compares the device's shelf sizes with the host's queues of point ids)
    monkeypatch.setattr(multi_nested_sampler, "_DEBUG_SHELVES", True)
    g = load_trace(case)
    ndata, nlive = int(g["ndata"]), int(g["nlive"])
    use_graph = bool(g.get("use_graph", 0))
    data = (gen.horns if "horns" in case else gen.nothing)(ndata)
    problem = sample.GaussLineProblem(data["x"], data["y"])          # HIP backend
    sampler = sample.build_sampler(problem, nlive_points=nlive, nsuperset_draws=int(g["nsuperset_draws"]),
                                   use_graph=use_graph, seed=1, batched=True, fused=fused)
    rec = Recorder(sampler)
    with np.errstate(all="ignore"):
        results = multi_nested_integrator(tolerance=0.5, multi_sampler=rec, min_samples=0,
                                          max_samples=int(g["max_samples"]))
    rng_probe = np.random.uniform()
    if fused:
        assert type(sampler.joint).__name__ == "GaussJointState"
        assert (sampler._dgroups is not None) == use_graph
    check_bookkeeping(g, sampler, rec, results)
    assert rng_probe == float(g["rng_probe"])
    check_floats(g, rec, results, rtol=1e-12)
    assert rel_err(results["logZ"], g["logZ"]) < 1e-6
    assert np.max(np.abs(results["logZ"] - g["logZ"])) < 1e-9


def test_region_handle_bit_exact(hip, nb, oracle):
    """Resident members (mdns_region_*): same numbers as the one-shot entry points."""
    rng = np.random.RandomState(21)
    for K, ndim in ((100, 3), (777, 5), (3000, 3)):
        pts = rng.uniform(size=(K, ndim))
        ms = nb.MemberSet(pts)
        chosen = np.zeros((K, 10))
        for b in range(10):
            chosen[rng.choice(np.arange(K), size=K, replace=True), b] = 1.
        r = ms.bootstrap_radius(chosen)
        assert r == oracle.bootstrapped_maxdistance(pts, chosen)
        for M in (1, 1000, 4097):
            cand = rng.uniform(-0.1, 1.1, size=(M, ndim))
            want = oracle.count_within_distance_of(pts, r, cand).astype(int)
            assert np.array_equal(ms.count(cand), want)
            assert np.array_equal(ms.any(cand), want > 0)
        ms.set_radius(0.5 * r)
        cand = rng.uniform(size=(500, ndim))
        assert np.array_equal(ms.count(cand), oracle.count_within_distance_of(pts, 0.5 * r, cand).astype(int))
        ms.close()
    # a region without a radius refuses to count
    ms = nb.MemberSet(rng.uniform(size=(10, 3)))
    with pytest.raises(_lib.MdnsError):
        ms.count(rng.uniform(size=(5, 3)))


def test_dropin_cmuselike_and_cneighbors_shims(golden, oracle):
    """The other two shim libraries, bound exactly as musefuse.py:509-517 and
    clustering/neighbors.py:100-166 bind the reference's."""
    nd_ = np.ctypeslib.ndpointer
    f2 = nd_(dtype=np.float64, ndim=2, flags='C_CONTIGUOUS')
    f1 = nd_(dtype=np.float64, ndim=1, flags='C_CONTIGUOUS')
    lm = C.CDLL(os.path.join(_lib.DROPIN_DIR, "cmuselike.so"))
    lm.like.argtypes = [f2, f2, f1, nd_(dtype=np.bool_, ndim=1, flags='C_CONTIGUOUS'), C.c_int, C.c_int, f1]
    yy, vv = golden["k2_y"], golden["k2_v"]
    nx, nd = yy.shape
    m = np.ascontiguousarray(golden["k2_masks"][1])
    Lout = np.zeros(nd)
    assert lm.like(yy, vv, np.ascontiguousarray(golden["k2_ypred"][0]), m, nd, nx, Lout) == 0
    assert rel_err(Lout[m], golden["k2_out1"][0][m]) < 1e-11 and np.all(Lout[~m] == 0)

    ln = C.CDLL(os.path.join(_lib.DROPIN_DIR, "cneighbors.so"))
    ln.most_distant_nearest_neighbor.argtypes = [f2, C.c_int, C.c_int]
    ln.most_distant_nearest_neighbor.restype = C.c_double
    ln.is_within_distance_of.argtypes = [f2, C.c_int, C.c_int, C.c_double, f1]
    ln.is_within_distance_of.restype = C.c_int
    ln.count_within_distance_of.argtypes = [f2, C.c_int, C.c_int, C.c_double, f2, C.c_int, f1, C.c_int]
    ln.bootstrapped_maxdistance.argtypes = [f2, C.c_int, C.c_int, f2, C.c_int]
    ln.bootstrapped_maxdistance.restype = C.c_double
    pts = golden["k6_d3_pts"]
    assert ln.most_distant_nearest_neighbor(pts, len(pts), 3) == float(golden["k6_d3_nn"])
    chosen = np.ascontiguousarray(golden["k6_d3_chosen"][0])
    assert ln.bootstrapped_maxdistance(pts, len(pts), 3, chosen, 10) == golden["k6_d3_radius"][0]
    mem, cand, r = golden["k3_d3_members"], golden["k3_d3_cands"], float(golden["k3_d3_r"])
    counts = np.zeros(len(cand))
    assert ln.count_within_distance_of(mem, len(mem), 3, r, cand, len(cand), counts, 0) == 0
    assert np.array_equal(counts, golden["k3_d3_count0"])
    assert ln.is_within_distance_of(mem, len(mem), 3, r, np.ascontiguousarray(cand[0])) == int(golden["k3_d3_any"][0])


def test_registered_spectra_are_reused(hip, golden):
    """mdns_register_spectra: later drop-in like() calls with the same pointer use the resident
    copy (and follow the x passed with each call, as clike.c:35 takes it per call)."""
    x, y = golden["k1_horns_x"], golden["k1_horns_y"]
    nx, nd = y.shape
    m = np.ones(nd, dtype=np.bool_)
    p = golden["k1_horns_params"][2]
    assert hip.mdns_register_spectra(_lib.ptr(y), None, nd, nx) == 0
    try:
        for xs in (x, x + 3.0):
            out = np.zeros(nd)
            assert hip.mdns_gauss_like(_lib.ptr(xs), _lib.ptr(y), nd, nx, p[0], p[1], 10 ** p[2], 0.01,
                                       _lib.ptr(m), _lib.ptr(out)) == 0
            ypred = p[0] * np.exp(-0.5 * ((p[1] - xs) / 10 ** p[2]) ** 2)
            want = (((ypred.reshape((-1, 1)) - y) / 0.01) ** 2).sum(axis=0)
            assert rel_err(out, want) < 1e-12
    finally:
        assert hip.mdns_unregister_spectra(_lib.ptr(y)) == 0
    assert hip.mdns_unregister_spectra(_lib.ptr(y)) != 0


def test_c4_size_single_gpu():
    """BASELINE.json configs[3]'s data -- gensimple_horns, 100 000 spectra -- on one GPU: sizes,
    indexing and the XCD tiling at a grid 10x larger than the bench's; checked by properties
    (the oracle checks one GPU's 12 500-spectra shard of it in the next test)."""
    from massivedatans_amd.like import GaussLineSpectra
    d = gen.horns(100000)
    sp = GaussLineSpectra(d["x"], d["y"])
    rng = np.random.RandomState(8)
    params = np.column_stack([rng.uniform(0.01, 1, 40), rng.uniform(400, 800, 40), 10 ** rng.uniform(0, 2, 40)])
    full = sp.loglike_batch(params)                                  # lane kernel
    assert full.shape == (40, 100000) and np.all(np.isfinite(full))
    one = sp.loglike_batch(params[3:4])[0]                           # row kernel
    assert rel_err(one, full[3]) < 1e-13
    cols = rng.choice(100000, size=257, replace=False)
    ypred = params[3, 0] * np.exp(-0.5 * ((params[3, 1] - d["x"]) / params[3, 2]) ** 2)
    want = -0.5 * (((ypred.reshape((-1, 1)) - d["y"][:, cols]) / 0.01) ** 2).sum(axis=0)
    assert rel_err(full[3, cols], want) < 1e-12
    m = np.zeros(100000, bool)
    m[cols] = True
    assert np.array_equal(sp.loglike_batch(params, m), full[:, m])   # 40 candidates, sparse: lane kernel + gather
    sp.close()


def test_c4_shard_against_the_oracle(oracle):
    """One GPU's share of configs[3]: spectra 37 500..50 000 of gensimple_horns(100 000), i.e. what
    rank 3 of 8 holds (parallel.shard_range), against the CPU oracle on the same columns -- the
    batched lane kernel, the one-candidate row kernel and a sparse selection; plus the joint
    state on that shard: a draw chunk decided on the device against the oracle's likelihoods."""
    from massivedatans_amd import jointstate, parallel, sample
    from massivedatans_amd.like import GaussLineSpectra
    d = gen.horns(100000)
    lo, hi = parallel.shard_range(100000, 3, 8)
    assert (lo, hi) == (37500, 50000)
    y = np.ascontiguousarray(d["y"][:, lo:hi])
    sp = GaussLineSpectra(d["x"], y)
    rng = np.random.RandomState(12)
    params = np.column_stack([rng.uniform(0.01, 1, 24), rng.uniform(400, 800, 24), 10 ** rng.uniform(0, 2, 24)])
    mask = np.ones(hi - lo, dtype=np.bool_)
    got = sp.loglike_batch(params)
    for b in (0, 7, 23):
        want = -0.5 * oracle.gauss_like(d["x"], y, params[b, 0], params[b, 1], params[b, 2], 0.01, mask)
        assert rel_err(got[b], want) < RTOL_L
    assert rel_err(sp.loglike_batch(params[5:6])[0], got[5]) < 1e-13
    sel = rng.uniform(size=hi - lo) < 0.02
    want = -0.5 * oracle.gauss_like(d["x"], y, params[2, 0], params[2, 1], params[2, 2], 0.01, sel)
    assert rel_err(sp.loglike_batch(params[:3], sel)[2], want) < RTOL_L
    # a draw chunk on the shard: thresholds = lowest of 20 live points per spectrum
    nlive = 20
    js = jointstate.GaussJointState(sp, nlive, sample.kernel_params)
    xs0 = sample.priortransform_batch(rng.uniform(size=(nlive, 3)))
    js.init(xs0)
    Lmin, arg, _ = js.prepare()
    live = np.array([-0.5 * oracle.gauss_like(d["x"], y, p[0], p[1], p[2], 0.01, mask) for p in sample.kernel_params(xs0)])
    assert np.array_equal(arg, live.argmin(axis=0)) and rel_err(Lmin, live.min(axis=0)) < RTOL_L
    cube = rng.uniform(size=(64, 3))
    xs = sample.priortransform_batch(cube)
    idx, Lrow, beats, n = js.draw(xs, None)
    Ls = np.array([-0.5 * oracle.gauss_like(d["x"], y, p[0], p[1], p[2], 0.01, mask) for p in sample.kernel_params(xs[:n])])
    ok = (Ls > live.min(axis=0)).any(axis=1)
    assert idx == (int(np.argmax(ok)) if ok.any() else -1)
    if idx >= 0:
        assert rel_err(Lrow, Ls[idx]) < RTOL_L and np.array_equal(beats, Ls[idx] > live.min(axis=0))
    js.close()
    sp.close()


@pytest.mark.parametrize("nd,nx,B", [(5000, 37, 40), (4100, 200, 50), (2500, 9, 80)])
def test_k1_lane_kernel_vs_oracle(oracle, nd, nx, B):
    """Shapes that dispatch to the lane-per-spectrum kernel (M*B >= 150 000), including channel
    counts that are not multiples of the pipeline depth and spectra counts that are not multiples
    of 64, full and gathered (dense mask) selections."""
    from massivedatans_amd.like import GaussLineSpectra
    rng = np.random.RandomState(nd + nx)
    x = np.sort(rng.uniform(400, 800, nx))
    y = np.ascontiguousarray(rng.normal(0, 0.05, size=(nx, nd)))
    sp = GaussLineSpectra(x, y, noise_level=0.01)
    params = np.column_stack([rng.uniform(0.01, 1, B), rng.uniform(400, 800, B), 10 ** rng.uniform(0, 2, B)])
    full_mask = np.ones(nd, bool)
    dense = rng.uniform(size=nd) < 0.8
    for mask in (full_mask, dense):
        assert mask.sum() * B >= 150000
        got = sp.loglike_batch(params, mask)
        for b in (0, B // 2, B - 1):
            want = -0.5 * oracle.gauss_like(x, y, params[b, 0], params[b, 1], params[b, 2], 0.01, mask)
            assert rel_err(got[b], want) < RTOL_L
    sp.close()


def test_device_side_threshold_matches_host(nb, oracle):
    """After mdns_region_bootstrap_radius* the membership threshold lives on the device
    (radius_and_threshold, run by the last workgroup of the bootstrap kernel); it must select
    exactly the points the reference's sqrt(d) < r selects, including candidates placed AT distance r from a member."""
    rng = np.random.RandomState(77)
    for trial in range(60):
        K, ndim = int(rng.randint(5, 40)), int(rng.randint(1, 4))
        pts = rng.uniform(size=(K, ndim)) * 10 ** rng.uniform(-3, 3)
        chosen = np.zeros((K, 10))
        for b in range(10):
            chosen[rng.choice(np.arange(K), size=K, replace=True), b] = 1.
        ms = nb.MemberSet(pts)
        r = ms.bootstrap_radius(chosen)                     # device radius + device threshold
        assert r == oracle.bootstrapped_maxdistance(pts, chosen)
        cand = [pts[i] + r * np.eye(ndim)[k] * s for i in range(min(K, 6)) for k in range(ndim) for s in (1, -1)]
        cand += [pts[0] + np.nextafter(r, 0) * np.eye(ndim)[0], pts[0] + np.nextafter(r, np.inf) * np.eye(ndim)[0]]
        cand = np.array(cand + list(rng.uniform(pts.min(), pts.max(), size=(50, ndim))))
        want = oracle.count_within_distance_of(pts, r, cand).astype(int)
        assert np.array_equal(ms.count(cand), want)         # threshold read from device memory
        ms.set_radius(r)
        assert np.array_equal(ms.count(cand), want)         # threshold computed on the host
        ms.close()


@pytest.mark.parametrize("nd,nx", [(601, 96), (530, 700), (515, 4096)])
def test_k2_two_rows_per_workgroup_vs_oracle(oracle, nd, nx):
    """Enough spectra and candidates (M >= 512, B >= 4) to take k_muse_rows2, with an odd row
    count, against the oracle; plus a sparse selection that falls back to k_muse_rows."""
    from massivedatans_amd.like import MuseSpectra
    cube = gen.muse_like(nd, nx=nx)
    sp = MuseSpectra(cube["x"], cube["y"], cube["v"])
    rng = np.random.RandomState(nx)
    B = 5
    pars = np.column_stack([rng.uniform(-0.5, 0.5, B), rng.uniform(0, 0.02, B), rng.uniform(-0.1, 0.2, B),
                            rng.uniform(0.5, 1.5, B), rng.uniform(0.5, 1.5, B)])
    ypred = np.array([gen.muse_template(cube["x"], p) for p in pars])
    for mask in (np.ones(nd, bool), rng.uniform(size=nd) < 0.2):
        got = sp.loglike_batch(ypred, mask)
        for b in range(B):
            want = oracle.muse_like(cube["y"], cube["v"], np.ascontiguousarray(ypred[b]), mask)[mask]
            assert rel_err(got[b], want) < 1e-11
    sp.close()


@pytest.mark.parametrize("nd,nx,B", [(12500, 200, 200), (777, 33, 300), (3300, 24, 1500), (10000, 200, 128)])
def test_k1_lane_kernel_every_entry(oracle, nd, nx, B):
    """Every entry of L[B, M] of the lane kernel, for shapes that exercise its placement logic:
    12 500 spectra (49 quads: XCDs with unequal and empty shares, 7 items per CU), candidate
    counts that are not multiples of the tile, the 16-candidate tile (1500 x 3300) and the
    8-candidate tile at B = 128.  A work item that no workgroup picked up would leave its block
    of the output untouched, so the output buffer is poisoned first."""
    from massivedatans_amd.like import GaussLineSpectra
    rng = np.random.RandomState(nd + nx + B)
    x = np.sort(rng.uniform(400, 800, nx))
    y = np.ascontiguousarray(rng.normal(0, 0.05, size=(nx, nd)))
    sp = GaussLineSpectra(x, y, noise_level=0.01)
    params = np.column_stack([rng.uniform(0.01, 1, B), rng.uniform(400, 800, B), 10 ** rng.uniform(0, 2, B)])
    mask = np.ones(nd, bool)
    lib = _lib.require_device()
    got = np.full((B, nd), np.nan)
    d_L = lib.mdns_dev_alloc(got.nbytes)
    d_p = lib.mdns_dev_alloc(params.nbytes)
    params = np.ascontiguousarray(params)
    _lib.check(lib.mdns_h2d(d_L, _lib.ptr(got), got.nbytes), "h2d")          # poison
    _lib.check(lib.mdns_h2d(d_p, _lib.ptr(params), params.nbytes), "h2d")
    _lib.check(lib.mdns_gauss_loglike_batch_dev(sp.handle, d_p, B, 0.01, None, nd, d_L), "K1")
    _lib.check(lib.mdns_d2h(_lib.ptr(got), d_L, got.nbytes), "d2h")
    lib.mdns_dev_free(d_L)
    lib.mdns_dev_free(d_p)
    assert np.isfinite(got).all(), "work items left untouched: %d entries" % (~np.isfinite(got)).sum()
    assert (lib.mdns_profile_kernel(0) or b"").startswith(b"k_gauss_cols")
    for b0 in range(0, B, 50):
        p = params[b0:b0 + 50]
        model = p[:, 0, None] * np.exp(-0.5 * ((p[:, 1, None] - x[None, :]) / p[:, 2, None]) ** 2)   # [b, nx]
        want = np.empty((len(p), nd))
        for i in range(len(p)):
            want[i] = -0.5 * (((model[i][:, None] - y) / 0.01) ** 2).sum(axis=0)
        assert rel_err(got[b0:b0 + 50], want) < 1e-11
    b = B - 1
    want = -0.5 * oracle.gauss_like(x, y, params[b, 0], params[b, 1], params[b, 2], 0.01, mask)
    assert rel_err(got[b], want) < RTOL_L
    sp.close()


@pytest.mark.parametrize("nd,nx,B", [(7, 700, 33), (45, 4096, 64), (1, 1500, 9)])
def test_k2_few_rows_many_candidates(oracle, nd, nx, B):
    """Few spectra and many templates: the candidates are split over grid.y (odd counts, a
    single row, full and partial selections).  Every entry against the oracle."""
    from massivedatans_amd.like import MuseSpectra
    cube = gen.muse_like(nd, nx=nx)
    sp = MuseSpectra(cube["x"], cube["y"], cube["v"])
    rng = np.random.RandomState(nx + B)
    pars = np.column_stack([rng.uniform(-0.5, 0.5, B), rng.uniform(0, 0.02, B), rng.uniform(-0.1, 0.2, B),
                            rng.uniform(0.5, 1.5, B), rng.uniform(0.5, 1.5, B)])
    ypred = np.array([gen.muse_template(cube["x"], p) for p in pars])
    masks = [np.ones(nd, bool)]
    if nd > 2:
        masks.append(np.arange(nd) % 3 != 1)
    for mask in masks:
        got = sp.loglike_batch(ypred, mask)
        assert got.shape == (B, int(mask.sum()))
        for b in range(B):
            want = oracle.muse_like(cube["y"], cube["v"], np.ascontiguousarray(ypred[b]), mask)[mask]
            assert rel_err(got[b], want) < 1e-11
    sp.close()


@pytest.mark.parametrize("nd,frac,B", [(10000, 0.1, 256), (10000, 0.011, 40), (3000, 0.5, 130), (5000, 0.0003, 64)])
def test_k1_sparse_selection_is_compacted_exactly(nd, frac, B):
    """Selections that the lane kernel scores from a compact replica of the selected spectra
    (many candidates or a sparse mask): the same numbers, bit for bit, as the corresponding
    columns of the full-mask result of the same kernel."""
    from massivedatans_amd.like import GaussLineSpectra
    d = gen.horns(nd)
    sp = GaussLineSpectra(d["x"], d["y"], noise_level=0.01)
    rng = np.random.RandomState(B)
    params = np.column_stack([rng.uniform(0.01, 1, B), rng.uniform(400, 800, B), 10 ** rng.uniform(0, 2, B)])
    full = sp.loglike_batch(params, np.ones(nd, bool))
    mask = rng.uniform(size=nd) < frac
    mask[rng.randint(nd)] = True
    got = sp.loglike_batch(params, mask)
    assert got.shape == (B, int(mask.sum()))
    assert np.array_equal(got, full[:, mask])
    sp.close()


def test_c5_share_size_properties():
    """Config C5's share of one of 8 GPUs (6 250 spectra x 4096 channels, 410 MB of y and 1/v on
    the device): sizes and indexing at full scale, checked by properties -- the batched kernels
    (two rows per workgroup) against the one-template kernel, a selection against the same
    columns of the full result, templates evaluated on the device against uploaded ones, and
    the defining formula (cmuselike.c:45-64) in numpy on a sample of spectra."""
    from massivedatans_amd.like import MuseSpectra
    nd, nx, B = 6250, 4096, 6
    cube = gen.muse_like(nd, nx=nx)
    sp = MuseSpectra(cube["x"], cube["y"], cube["v"])
    rng = np.random.RandomState(5)
    pars = np.column_stack([rng.uniform(-0.5, 0.5, B), rng.uniform(0, 0.02, B), rng.uniform(-0.1, 0.2, B),
                            rng.uniform(0.5, 1.5, B), rng.uniform(0.5, 1.5, B)])
    ypred = np.array([gen.muse_template(cube["x"], p) for p in pars])
    full = sp.loglike_batch(ypred)                               # two rows per workgroup
    assert full.shape == (B, nd) and np.all(np.isfinite(full)) and np.all(full <= 0)
    one = sp.loglike_batch(ypred[2:3])[0]                        # one-template kernel
    assert rel_err(one, full[2]) < 1e-12
    assert rel_err(sp.loglike_batch_lines(pars), full) < 1e-9    # device-side templates
    mask = rng.uniform(size=nd) < 0.3
    assert rel_err(sp.loglike_batch(ypred, mask), full[:, mask]) < 1e-12
    for i in rng.choice(nd, size=40, replace=False):
        y, v = cube["y"][:, i], cube["v"][:, i]
        for b in (0, B - 1):
            m = ypred[b]
            s = (y * m / v).sum() / (1e-10 + (m * m / v).sum())
            want = -0.5 * (((y - s * m) ** 2) / v).sum()
            assert abs(full[b, i] - want) <= 1e-10 * abs(want)
    sp.close()


def test_packed_bootstrap_choice_gives_the_same_radius(nb, oracle):
    """mdns_region_bootstrap_radius_packed (bit masks built by draw_bootstrap_masks) against the
    f64 choice matrix of the reference's interface and the oracle: bit-exact radii, for pools on
    both sides of the fused-kernel limit and several round counts."""
    rng = np.random.RandomState(11)
    for K, ndim, B in [(7, 2, 10), (100, 3, 10), (400, 3, 10), (1500, 3, 10), (3000, 5, 16), (2500, 3, 3)]:
        pts = rng.uniform(size=(K, ndim))
        np.random.seed(K + B)
        masks = nb.draw_bootstrap_masks(K, B)
        chosen = nb.unpack_bootstrap_masks(masks, B)
        want = oracle.bootstrapped_maxdistance(np.ascontiguousarray(pts), np.ascontiguousarray(chosen))
        s = nb.MemberSet(pts)
        assert s.bootstrap_radius_packed(masks, B) == want
        assert s.bootstrap_radius(chosen) == want
        far = pts[:5] + want * 0.999 / np.sqrt(ndim)
        assert np.array_equal(s.count(far), oracle.count_within_distance_of(np.ascontiguousarray(pts), want, np.ascontiguousarray(far)))
        s.close()


@pytest.mark.parametrize("kind,ndata,cap", [("horns", 10000, 400), ("nothing", 10000, 400), ("horns", 10000, 700),
                                            ("horns", 100000, 300), ("horns", 100000, 450)])
def test_full_size_bookkeeping_matches_the_cpu_path(kind, ndata, cap):
    """BASELINE.json configs[1]/[2] at full size (10 000 spectra, 100 live points), first 400
    iterations (and 700 for horns: 265 425 draws, 40 minutes on the CPU path), and configs[3]'s
    data set whole (gensimple_horns, 100 000 spectra, 300 and 450 iterations) on the one GPU, against the
    same run of the host orchestration on the CPU oracle
    backends (tests/golden/bookkeeping_c2.json from oracle/make_bookkeeping_hash.py; that pair is
    pinned bit for bit against the reference on the small traces): same number of draws, the
    pile of accepted points byte for byte -- i.e. every accept decision and every RNG draw
    coincided over 12 117 (5 261, 265 425, 2 992, 31 982) likelihood calls -- and evidences within 1e-9."""
    import hashlib
    import json
    from massivedatans_amd import sample
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = json.load(open(os.path.join(root, "tests", "golden", "bookkeeping_c2.json")))["%s_%d_100_%d" % (kind, ndata, cap)]
    data = (gen.horns if kind == "horns" else gen.nothing)(ndata)
    with np.errstate(all="ignore"):
        results, sampler, _, _ = sample.run(data["x"], data["y"], nlive_points=100, max_samples=cap, use_graph=False)
    assert sampler.ndraws == want["ndraws"]
    assert len(sampler.pointpile) == want["npoints"]
    got = hashlib.sha256(np.ascontiguousarray(sampler.pointpile, dtype=np.float64).tobytes()).hexdigest()
    assert got == want["pointpile_sha256"]
    assert np.max(np.abs(results["logZ"][:5] - np.array(want["logZ_first5"]))) < 1e-9


def test_c3_is_the_low_acceptance_stress_it_is_meant_to_be():
    """BASELINE.json configs[2] (gennothing: no signal anywhere) exists to stress the RadFriends
    proposals: the likelihood surfaces of the data sets disagree, so regions stay large and most
    proposals are rejected.  On the first 400 iterations: the membership kernel (K3) sees many
    more proposals than ever become candidates, and a constrained draw needs several tries --
    both worse than on the horns data of configs[1], where a line pulls the data sets together."""
    from massivedatans_amd import sample
    from massivedatans_amd.clustering import neighbors
    stats = {}
    for kind in ("nothing", "horns"):
        data = (gen.nothing if kind == "nothing" else gen.horns)(10000)
        seen = {"proposed": 0, "inside": 0}
        orig_count, orig_any = neighbors.MemberSet.count, neighbors.MemberSet.any

        def count(self, points, _o=orig_count, _s=seen):
            c = _o(self, points)
            _s["proposed"] += len(c)
            _s["inside"] += int((c > 0).sum())
            return c

        neighbors.MemberSet.count = count
        neighbors.MemberSet.any = lambda self, points: count(self, points) > 0
        try:
            with np.errstate(all="ignore"):
                results, sampler, _, _ = sample.run(data["x"], data["y"], nlive_points=100, max_samples=400, use_graph=False)
        finally:
            neighbors.MemberSet.count, neighbors.MemberSet.any = orig_count, orig_any
        if sampler.native is not None:           # the draws ran in the library: its own counters
            t = sampler.native.stats()
            assert t["draws"] == sampler.ndraw_calls and t["tries"] == sampler.ndraws - 100
            seen = {"proposed": t["proposals"], "inside": t["inside"]}
        stats[kind] = dict(tries_per_draw=(sampler.ndraws - 100) / sampler.ndraw_calls,
                           k3_proposals_per_try=seen["proposed"] / (sampler.ndraws - 100),
                           inside_fraction=seen["inside"] / seen["proposed"])
        if sampler.joint is not None:
            sampler.joint.close()
    n, h = stats["nothing"], stats["horns"]
    assert n["tries_per_draw"] > 1.0 and n["k3_proposals_per_try"] > 1.0
    assert 0.0 < n["inside_fraction"] < 1.0
    print("RadFriends stress, first 400 iterations:", stats)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("kind", ["nothing", "horns"])
def test_full_run_matches_the_cpu_path(kind, graph):
    """BASELINE.json configs[2] (10 000 no-signal spectra) / configs[1] (horns), 100 live points,
    TO TERMINATION on the GPU against the same complete run on the CPU oracle backends
    (tests/golden/full_c3.npz / full_c2.npz from oracle/make_full_run.py): same iterations and
    draws, the pile of accepted points byte for byte, and the evidences of all 10 000 data sets
    within 1e-9 (relative bar of BASELINE.json: 1e-6).  The horns run takes a minute and a half
    on the GPU (2.3 hours on the CPU path); MDNS_SKIP_LONG_TESTS=1 leaves it out.  ``graph``: the
    reference's default grouping -- connected components, on the device here, by the host
    implementation in the fixture's run (full_c3_graph.npz / full_c2_graph.npz)."""
    import hashlib
    from massivedatans_amd import sample
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", ("full_c3" if kind == "nothing" else "full_c2") + ("_graph" if graph else "") + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated (oracle/make_full_run.py %s%s)" % (kind, " graph" if graph else ""))
    if kind == "horns" and os.environ.get("MDNS_SKIP_LONG_TESTS") == "1":
        pytest.skip("MDNS_SKIP_LONG_TESTS=1")
    with np.load(path) as f:
        want = {k: f[k] for k in f.files}
    data = (gen.nothing if kind == "nothing" else gen.horns)(10000)
    with np.errstate(all="ignore"):
        results, sampler, _, _ = sample.run(data["x"], data["y"], nlive_points=100, max_samples=0, use_graph=graph)
    if graph:
        assert sampler._dgroups is not None, "the components did not come from the device"
    assert results["nsamples"] == int(want["iterations"])
    assert sampler.ndraws == int(want["ndraws"])
    assert len(sampler.pointpile) == int(want["npoints"])
    got = hashlib.sha256(np.ascontiguousarray(sampler.pointpile, dtype=np.float64).tobytes()).hexdigest()
    assert got == str(want["pointpile_sha256"])
    assert np.max(np.abs(results["logZ"] - want["logZ"])) < 1e-9
    assert np.max(np.abs(results["logZ"] - want["logZ"]) / np.abs(want["logZ"])) < 1e-6
    assert np.allclose(results["logZerr"], want["logZerr"], rtol=1e-6, atol=1e-9)


def test_muse_multi_loglikelihood_with_jitter(oracle):
    """The K2 caller on the GPU (like.MuseSpectra.multi_loglikelihood, musefuse.py:520-535): values
    against the oracle + the reference's N(0, 1e-5) tie-breaker, and the global RNG stream left
    where the reference leaves it (one normal deviate per selected spectrum, none for a
    template without stars)."""
    from massivedatans_amd.like import MuseSpectra
    from test_sampler_units import _check_muse_jitter, _muse_case
    for nd, nx in ((23, 61), (300, 4096)):
        x, y, v, ypred, mask = _muse_case(seed=nd, nd=nd, nx=nx)
        spectra = MuseSpectra(x, y, v)
        _check_muse_jitter(spectra, oracle, y, v, ypred, mask, rtol=1e-11)
        spectra.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,ndim,nboot", [(1024, 3, 10), (1500, 1, 10), (2047, 2, 10), (5000, 3, 10), (4100, 4, 16), (3000, 5, 7),
                                          (16384, 3, 10), (9000, 3, 10)])
def test_k6_in_morton_order_gives_the_all_pairs_radius(K, ndim, nboot, oracle, monkeypatch):
    """K6 with the pool sorted along a Morton curve and far tiles culled (csrc/mdns_k6sort.hip) against the
    all-pairs kernel (the default) and the oracle, bit for bit -- on point sets built to
    hurt a spatial scheme: a dense core inside a thin halo, duplicated points, a degenerate axis, and the
    pool's point 0 (which the reference never lets contribute, cneighbors.c:162) placed far outside."""
    import subprocess, sys, json
    from massivedatans_amd.clustering import neighbors
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(K + ndim)
    core = rng.normal(0.5, 0.002, size=(K // 2, ndim))
    halo = rng.uniform(size=(K - K // 2, ndim))
    pts = np.vstack((core, halo))
    rng.shuffle(pts)
    pts[7] = pts[3]                                       # duplicates: distance 0 pairs
    pts[K // 3:K // 3 + 50] = pts[K // 3]
    if ndim > 1:
        pts[:, ndim - 1] = 0.25                           # a degenerate axis
    pts[0] = 7.0                                          # far away: would set the radius if it counted
    np.random.seed(K)
    masks = neighbors.draw_bootstrap_masks(K, nboot)
    chosen = neighbors.unpack_bootstrap_masks(masks, nboot)
    want = oracle.bootstrapped_maxdistance(np.ascontiguousarray(pts), np.ascontiguousarray(chosen))
    s = neighbors.MemberSet(pts)
    got = s.bootstrap_radius_packed(masks, nboot)          # the default path: all pairs
    s.close()
    assert got == want
    # the same through the sorted form, in a fresh process (the path is chosen once per process)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from massivedatans_amd.clustering import neighbors; "
            "d = np.load(sys.argv[1]); s = neighbors.MemberSet(d['pts']); print(repr(s.bootstrap_radius_packed(d['masks'], int(d['nboot']))))" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "k6.npz")
        np.savez(path, pts=pts, masks=masks, nboot=nboot)
        out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, MDNS_K6_PATH="sorted"))
    assert out.returncode == 0, out.stderr[-1500:]
    assert float(out.stdout.strip().splitlines()[-1]) == want
