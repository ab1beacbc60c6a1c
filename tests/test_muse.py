"""BASELINE.json configs[4] as a CONSTRAINED DRAW: the likelihood of musefuse.py:520-535 -- host or
device template, the scale-marginalised chi^2 of cmuselike.c:45-64, N(0, 1e-5) noise from the
global random stream on every evaluation -- behind the sampler, through the classic Python
constrainer and through the native one (csrc/host_constrainer.cpp), against traces recorded from
the REFERENCE's own sampler / integrator / constrainers driven with the same problem definition
(oracle/make_trace.py, cases muse*)."""
import os
import sys

import numpy as np
import pytest

from massivedatans_amd import _lib, gen, musefuse
from oracle_backend import OracleMuseSpectra, patch_neighbors
from tracing import Recorder, check_bookkeeping, check_floats, load_trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(g, backend, fused, native, jitter=True):
    from massivedatans_amd import sample
    from massivedatans_amd.multi_nested_integrator import multi_nested_integrator
    ndata = int(g["ndata"])
    data = gen.muse_like(ndata, int(g["nx"]))
    problem = musefuse.MuseProblem(data["x"], data["y"], data["v"], backend=backend(data) if backend else None, jitter=jitter)
    sampler = sample.build_sampler(problem, nlive_points=int(g["nlive"]), nsuperset_draws=int(g["nsuperset_draws"]),
                                   use_graph=bool(g["use_graph"]), seed=1, batched=False, fused=fused, native=native)
    rec = Recorder(sampler)
    with np.errstate(all="ignore"):
        results = multi_nested_integrator(tolerance=0.5, multi_sampler=rec, min_samples=0, max_samples=int(g["max_samples"]))
    if sampler.native is not None:
        sampler.native.sync_gauss_to_numpy()
    return results, sampler, rec, np.random.uniform()


@pytest.mark.parametrize("case", ["muse6", "muse10_graph"])
@pytest.mark.parametrize("mode", ["single", "native", "native-block", "native-far-ahead", "native-not-ahead", "native-threads"])
def test_muse_trace_bit_exact(case, mode, oracle, monkeypatch):
    """``single``: one candidate per likelihood call, the noise from numpy.random.normal -- the
    reference's loop.  ``native``: whole chunks, the noise drawn in C from numpy's own Mersenne
    Twister (legacy polar Gaussian, its cached second deviate included) and the stream put back to
    where the accepted candidate's evaluation left it.  Both on the CPU oracle's cmuselike: integers,
    floats and the position of the random stream exactly the reference's."""
    g = load_trace(case)
    if mode == "native-block":
        # the noise drawn deviate by deviate and handed over as a [B, M] block (the round-3 form); "native"
        # takes the band form: the stream only advanced, exact deviates for undecided pairs and the
        # accepted candidate's row alone (csrc/host_constrainer.cpp, band_chunk)
        if case != "muse6":
            pytest.skip("the block form on the short trace only")
        monkeypatch.setenv("MDNS_JITTER_BAND", "0")
        mode = "native"
    ahead = None
    if mode == "native-threads":
        # the blocks of the noise stream made by two helper threads ahead of the scan (host_constrainer.cpp,
        # BlockProducer; opt-in): the blocks are a function of the stream alone, so the same trace
        if case != "muse6":
            pytest.skip("on the short trace only")
        monkeypatch.setenv("MDNS_BAND_THREADS", "1")
        monkeypatch.setenv("MDNS_BAND_READY_AFTER", "40")
        mode = "native"
    if mode in ("native-far-ahead", "native-not-ahead"):
        # the bounds of the candidates that follow a chunk in its batch are made while the chunk is scored
        # (band_chunk, BandLook): as far as the batch goes / not at all -- the same trace either way
        if case != "muse6":
            pytest.skip("on the short trace only")
        ahead = mode == "native-far-ahead"
        monkeypatch.setenv("MDNS_BAND_READY_AFTER", "100000" if ahead else "0")
        if not ahead:
            monkeypatch.setenv("MDNS_BAND_AHEAD", "0")
        mode = "native"
    if mode == "native":
        from massivedatans_amd import constrainer
        if not constrainer.available():
            pytest.skip("libmdns_host.so not built")
    if mode == "single" and case != "muse6" and os.environ.get("MDNS_LONG_TESTS") != "1":
        pytest.skip("one Python call per likelihood evaluation on the longer trace: with MDNS_LONG_TESTS=1")
    if mode == "single" and int(g["ndraws"]) > 400000:
        pytest.skip("one Python call per likelihood evaluation: hours")
    patch_neighbors(monkeypatch, oracle)
    results, sampler, rec, probe = _run(g, lambda d: OracleMuseSpectra(oracle, d["x"], d["y"], d["v"]),
                                        fused=(mode == "native"), native=(mode == "native"))
    assert (sampler.native is not None) == (mode == "native")
    check_bookkeeping(g, sampler, rec, results)
    check_floats(g, rec, results, rtol=0)
    assert probe == float(g["rng_probe"])
    if mode == "native":
        st = sampler.native.stats()
        assert (st["band_pairs"] + st["band_replays"] > 0) == (os.environ.get("MDNS_JITTER_BAND", "1") != "0") or st["band_pairs"] == 0
        if os.environ.get("MDNS_JITTER_BAND", "1") != "0":
            assert (st["band_ahead"] > 0) == (ahead is not False), st


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["muse6", "muse10_graph"])
def test_muse_on_the_gpu_against_the_reference_trace(case):
    """The same analysis with the K2 joint state on the GPU (templates evaluated on the device,
    k_muse_rows into a dense block, noise added and accept test there, commit, mailbox) under the
    native constrainer: the device's likelihoods differ from cmuselike's in the last bits, so the
    integer bookkeeping is expected -- not guaranteed -- to coincide; the evidences must agree within
    BASELINE's 1e-6 and the random stream must end where the reference's ends."""
    g = load_trace(case)
    results, sampler, rec, probe = _run(g, None, fused=True, native=True)
    assert type(sampler.joint).__name__ == "MuseJointState" and sampler.native is not None
    check_bookkeeping(g, sampler, rec, results)
    assert probe == float(g["rng_probe"])
    check_floats(g, rec, results, rtol=1e-9)
    assert np.max(np.abs(results["logZ"] - g["logZ"]) / np.abs(g["logZ"])) < 1e-6


@pytest.mark.gpu
def test_muse_joint_state_against_its_numpy_statement(oracle):
    """mdns_backend_draw_chunk for spectra with variances against jointstate.HostJointState over the
    oracle's cmuselike: accept index, fill bits, thresholds and live matrix, with and without noise,
    full and sparse selections."""
    from massivedatans_amd import jointstate
    from massivedatans_amd.like import MuseSpectra
    rng = np.random.RandomState(12)
    ndata, nlive, nx = 300, 24, 777
    data = gen.muse_like(ndata, nx)
    spectra = MuseSpectra(data["x"], data["y"], data["v"])
    dev = jointstate.MuseJointState(spectra, nlive, shelf_cap=4)
    host = jointstate.HostJointState(musefuse._LinesScorer(OracleMuseSpectra(oracle, data["x"], data["y"], data["v"])), nlive, ndata,
                                     musefuse.kernel_params, nparams=5)
    xs0 = musefuse.priortransform_batch(rng.uniform(size=(nlive, 5)))
    noise0 = rng.normal(0, 1e-5, size=(nlive, ndata))
    dev.init(xs0, jitter=noise0)
    host.init(xs0, jitter=noise0)
    assert np.allclose(dev.live_matrix(), host.live_matrix(), rtol=1e-10, atol=0)
    accepted = 0
    for it in range(6):
        a, b = dev.prepare(), host.prepare()
        assert np.array_equal(a[1], b[1]) and np.allclose(a[0], b[0], rtol=1e-10)
        waiting = np.zeros(ndata, dtype=int)
        for attempt in range(200):
            if (waiting > 0).all():
                break
            empty = np.flatnonzero(waiting == 0)
            rows = None if attempt < 2 else np.sort(rng.choice(empty, size=rng.randint(1, len(empty) + 1), replace=False)).astype(np.int32)
            M = ndata if rows is None else len(rows)
            B = int(rng.choice([1, 3, 9]))
            params = musefuse.priortransform_batch(rng.uniform(size=(B, 5)))
            noise = rng.normal(0, 1e-5, size=(B, M)) if attempt % 2 == 0 else None
            ia, _, ba, _ = dev.draw_params(params, rows, jitter=noise)
            ib, _, bb, _ = host.draw_params(params, rows, jitter=noise)
            assert ia == ib, (it, attempt, ia, ib)
            if ia >= 0:
                accepted += 1
                assert np.array_equal(ba, bb)
                waiting[(np.arange(ndata) if rows is None else rows)[ba]] += 1
        assert (waiting > 0).all()
        ha, hn = dev.thresholds()
        hb, hm = host.thresholds()
        assert np.array_equal(hn, hm) and np.allclose(ha, hb, rtol=1e-10)
        dev.advance()
        host.advance()
        assert np.allclose(dev.live_matrix(), host.live_matrix(), rtol=1e-10, atol=0)
    assert accepted >= 6           # at least one per iteration (a superset draw may fill every shelf at once)
    dev.close()


@pytest.mark.gpu
def test_configs4_whole_on_one_gpu():
    """BASELINE.json configs[4] WHOLE -- 50 000 spectra x 4096 channels, 3.3 GB of spectra and inverse
    variances -- on one GPU, behind the sampler: properties that do not need the CPU path (hours).
    The initial live points' likelihoods equal the stand-alone batch call; a constrained draw over
    all 50 000 data sets and one over a sparse selection deliver points that beat exactly the data
    sets the fill bits name (re-scored with the batch call); thresholds are order statistics."""
    from massivedatans_amd import jointstate
    from massivedatans_amd.like import MuseSpectra
    ndata, nx, nlive = 50000, 4096, 12
    rng = np.random.RandomState(7)
    x = np.linspace(4750, 9350, nx)
    # (gen.muse_like draws 4 x 10^8 deviates one spectrum at a time: minutes; the same recipe in blocks)
    z = rng.uniform(0.0, 0.02, size=ndata)
    scale = 10 ** rng.uniform(-1, 1, size=ndata)
    y = np.empty((nx, ndata))
    v = np.empty((nx, ndata))
    for lo in range(0, ndata, 5000):
        hi = lo + 5000
        truth = np.stack([scale[i] * gen.muse_template(x, (0.0, z[i], 0.0, 1.0, 1.0)) for i in range(lo, hi)], axis=1)
        v[:, lo:hi] = rng.uniform(0.5, 2.0, size=(nx, hi - lo)) * gen.NOISE_LEVEL ** 2
        y[:, lo:hi] = truth + rng.normal(0, 1, size=(nx, hi - lo)) * np.sqrt(v[:, lo:hi])
    spectra = MuseSpectra(x, y, v)
    del y, v
    js = jointstate.MuseJointState(spectra, nlive)
    xs0 = musefuse.priortransform_batch(rng.uniform(size=(nlive, 5)))
    js.init(xs0)
    live = js.live_matrix()
    want = spectra.loglike_batch_lines(xs0)
    assert live.shape == (nlive, ndata) and np.array_equal(live, want)
    Lmin, arg, _ = js.prepare()
    assert np.array_equal(Lmin, live.min(axis=0)) and np.array_equal(arg, live.argmin(axis=0))
    for rows in (None, np.sort(rng.choice(ndata, size=1500, replace=False)).astype(np.int32)):
        thr_before, n_before = js.thresholds()
        idx = -1
        for attempt in range(50):
            cube = rng.uniform(size=(8, 5))
            params = musefuse.priortransform_batch(cube)
            idx, _, beats, _ = js.draw_params(params, rows)
            if idx >= 0:
                break
        assert idx >= 0, "no acceptable candidate in 400 proposals"
        sel = np.arange(ndata) if rows is None else rows
        # (re-scored by the stand-alone batch call: another kernel shape, last bits may differ)
        L = spectra.loglike_batch_lines(params[idx:idx + 1], None if rows is None else rows)[0]
        clear = np.abs(L - thr_before[sel]) > 1e-9 * np.abs(L)
        assert np.array_equal(beats[clear], (L > thr_before[sel])[clear]) and clear.mean() > 0.999
        thr_after, n_after = js.thresholds()
        assert np.array_equal(n_after[sel], n_before[sel] + beats)
        # a data set with an empty shelf that took the point in: its threshold is now the 2nd smallest
        # of its live likelihoods and the new one
        pick = np.flatnonzero(beats & (n_before[sel] == 0))[:200]
        took = sel[pick]
        second = np.sort(np.vstack([live[:, took], L[pick][None, :]]), axis=0)[1, :]
        assert len(took) > 0 and np.allclose(thr_after[took], second, rtol=1e-12, atol=0)
    js.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ndata,nx,B,sparse", [(1500, 700, 40, False), (2000, 4096, 64, False), (1300, 333, 17, True), (600, 1024, 9, True),
                                                (6250, 4096, 64, False), (6250, 4096, 57, True)])
def test_k2_matrix_core_filter_decides_like_the_exact_kernels(ndata, nx, B, sparse):
    """mdns_backend_draw_band through the matrix-core filter (csrc/mdns_k2gemm.hip) against the exact row
    kernels, with thresholds planted on the candidates' own likelihoods at relative distances from 1e-3 down
    to 0 on both sides (the last two shapes: one GPU's share of BASELINE configs[4], whole -- tiled operands -- and a
    sparse selection of it -- row-major operands): every candidate's status and the listed pairs (which only the exact
    kernels may produce) are the same; the filter handed the chunk over (it cannot settle a threshold 1e-13 away); and
    the commit keeps the exact kernel's row."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import k2_filter_bench as kb
    rng = np.random.RandomState(ndata + B)
    rows = np.sort(rng.choice(ndata, size=ndata * 2 // 3, replace=False)).astype(np.int32) if sparse else None
    results = {}
    for mode in (0, 1):
        sp, st, params, L, thr = kb.planted_state(ndata, nx, 6, B, 5, [-1e-3, -1e-9, -1e-13, 0.0, 1e-13, 1e-9, 1e-3])
        lib = st._lib
        s0 = kb.stats(lib)
        res = kb.band(st, params, rows, np.zeros(B), mode)
        s1 = kb.stats(lib)
        M = ndata if rows is None else len(rows)
        bits = np.zeros((M + 63) // 64, dtype=np.uint64)
        first = int(np.flatnonzero(res[0] == 1)[0])
        st._check(lib.mdns_backend_draw_band_commit(st._h, first, _lib.ptr(np.zeros(M)), _lib.ptr(bits)), "draw_band_commit")
        higher, shelf_n = st.thresholds()
        results[mode] = (res[:6], bits.copy(), higher.copy(), shelf_n.copy(), [b - a for a, b in zip(s0, s1)])
        lib.mdns_muse_filter_mode(-1)
        st.close(); sp.close()
    exact, filt = results[0], results[1]
    assert np.array_equal(exact[0][0], filt[0][0])                     # status per candidate
    assert exact[0][1] == filt[0][1] and exact[0][1] > 0               # listed pairs: some thresholds are too close to call
    for a, b in zip(exact[0][2:], filt[0][2:]):
        assert np.array_equal(a, b)
    assert np.array_equal(exact[1], filt[1]) and np.array_equal(exact[2], filt[2]) and np.array_equal(exact[3], filt[3])
    assert exact[4][0] == 0 and filt[4][0] == 1 and filt[4][1] == 1 and filt[4][2] == 0   # filtered once, handed over, commit from the exact block
    assert (exact[0][0] == 1).any() and (exact[1] != 0).any()


@pytest.mark.gpu
def test_k2_matrix_core_filter_settles_clear_chunks_alone():
    """Thresholds 1 % away from every candidate: the filter decides the whole chunk, nothing is scored
    again, and the commit makes the accepted candidate's exact row (shelves and thresholds as without it)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import k2_filter_bench as kb
    out = {}
    for mode in (0, 1):
        sp, st, params, L, thr = kb.planted_state(1800, 1200, 6, 48, 9, [-1e-2, 1e-2])
        lib = st._lib
        s0 = kb.stats(lib)
        res = kb.band(st, params, None, np.full(48, 5e-5), mode)
        bits = np.zeros((1800 + 63) // 64, dtype=np.uint64)
        first = int(np.flatnonzero(res[0] == 1)[0])
        jrow = np.random.RandomState(3).normal(0, 1e-5, size=1800)
        st._check(lib.mdns_backend_draw_band_commit(st._h, first, _lib.ptr(jrow), _lib.ptr(bits)), "draw_band_commit")
        higher, shelf_n = st.thresholds()
        s1 = kb.stats(lib)
        out[mode] = (res[0].copy(), res[1], bits, higher, shelf_n, [b - a for a, b in zip(s0, s1)])
        lib.mdns_muse_filter_mode(-1)
        st.close(); sp.close()
    for a, b in zip(out[0][:5], out[1][:5]):
        assert np.array_equal(a, b)
    assert out[0][1] == 0
    assert out[1][5][:3] == [1, 0, 1] and out[0][5][:3] == [0, 0, 0]
