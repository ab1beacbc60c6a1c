"""The constrained draw decided on the GPU (include/mdns.h Part 2b: mdns_joint_*)
against its numpy statement ``jointstate.HostJointState``.

The host statement is fed likelihoods from the SAME lane kernel (the public batch call, padded
to 32+ candidates so that the shape-based dispatch takes it), so thresholds, accept decisions,
likelihood rows and fill bits have to agree EXACTLY, not just within a tolerance; a second test
runs the statement on the CPU oracle and allows the 1e-12 the two arithmetics differ by.
"""
import numpy as np
import pytest

from massivedatans_amd import gen, jointstate, sample
from massivedatans_amd.like import GaussLineSpectra

pytestmark = pytest.mark.gpu


class LaneScorer(object):
    """``loglike_batch`` that always runs the lane kernel (B >= 32 forces it)."""

    def __init__(self, spectra):
        self.spectra, self.ndata = spectra, spectra.ndata

    def loglike_batch(self, params, data_mask=None):
        params = np.atleast_2d(params)
        B = len(params)
        if B < 32:
            params = np.vstack([params] + [params[-1:]] * (32 - B))
        return self.spectra.loglike_batch(params, data_mask)[:B]


def _drive(dev, host, ndata, rng, iterations, exact):
    """The same sequence of iterations on both states: prepare, a few draw chunks on random
    selections until every running data set has something waiting, advance."""
    nlive = dev.nlive
    running = np.arange(ndata)
    ndraws = 0
    for it in range(iterations):
        if it == iterations // 2 and ndata > 8:
            running = np.sort(rng.choice(ndata, size=max(3, ndata * 2 // 3), replace=False))   # cut_down
            dev.set_running(running)
            host.set_running(running)
        a, b = dev.prepare(), host.prepare()
        assert np.array_equal(a[1], b[1])
        assert np.array_equal(a[0], b[0]) if exact else np.allclose(a[0], b[0], rtol=1e-12)
        assert (a[2] is None) == (b[2] is None)
        if a[2] is not None:
            w = min(a[2].shape[1], b[2].shape[1])
            assert np.array_equal(a[2][:, :w], b[2][:, :w]) and not a[2][:, w:].any() and not b[2][:, w:].any()
        waiting = np.zeros(ndata, dtype=int)
        waiting[running] = host.thresholds()[1][running]
        passes = 0
        while (waiting[running] == 0).any():
            passes += 1
            assert passes < 400, "the candidates never filled every shelf"
            if passes <= 2:
                rows = running                                      # superset draw
            else:
                empty = running[waiting[running] == 0]
                rows = np.sort(rng.choice(empty, size=rng.randint(1, len(empty) + 1), replace=False))
            B = int(rng.choice([1, 3, 17, 64, 200]))
            cube = rng.uniform(size=(B, 3))
            if passes > 6:
                cube[:, 0] *= 0.05                                  # faint lines beat more thresholds
            xs = sample.priortransform_batch(cube)
            ha, hn = dev.thresholds()
            hb, hm = host.thresholds()
            assert np.array_equal(hn[running], hm[running])
            assert np.array_equal(ha[running], hb[running]) if exact else np.allclose(ha[running], hb[running], rtol=1e-12)
            sel = None if len(rows) == ndata else rows
            xs = xs[:dev.chunk_size(len(xs), len(rows), hint=int(rng.randint(1, 80)))]
            ia, La, ba, na = dev.draw(xs, sel)
            ib, Lb, bb, nb_ = host.draw(xs[:na], sel)
            assert ia == ib, (it, passes, ia, ib)
            if ia >= 0:
                ndraws += 1
                assert np.array_equal(ba, bb)
                if La is not None:                                  # (None: the state keeps the row to itself)
                    assert np.array_equal(La, Lb) if exact else np.allclose(La, Lb, rtol=1e-12)
                waiting[rows[ba]] += 1
        dev.advance()
        host.advance()
        la, lb = dev.live_matrix(), host.live_matrix()
        assert np.array_equal(la, lb) if exact else np.allclose(la, lb, rtol=1e-12)
    return ndraws


@pytest.mark.parametrize("fetch_rows", [True, False, "backend"])
@pytest.mark.parametrize("ndata,nlive,nx", [(1, 5, 200), (7, 9, 33), (100, 50, 200), (1000, 40, 200), (4100, 25, 64), (700, 30, 201),
                                           (90, 150, 48)])        # more live points than a data set's sixteen lanes hold in registers (128)
def test_joint_state_equals_its_numpy_statement(ndata, nlive, nx, fetch_rows):
    """fetch_rows=False: the outcome of a draw -- index, fill bits -- arrives in mapped host memory
    the commit kernel writes (mdns.h, mdns_joint_fetch), the likelihood row stays on the device;
    True copies the result buffer back.  "backend" is what a real run uses: the entry points a
    native constrainer calls (mdns_backend_draw_begin / _chunk) -- for selections of up to 4096
    spectra the chunk is two launches (csrc/mdns_chunk.hip: templates computed in the accept
    kernel, candidates and row ids read from mapped host memory, commit + mailbox in one
    workgroup); the shelf capacity of 4 also makes it grow the shelves itself."""
    rng = np.random.RandomState(ndata * 7 + nlive)
    data = gen.horns(ndata)
    x = np.linspace(400, 800, nx) if nx > 200 else data["x"][:nx]
    y = np.ascontiguousarray(np.vstack([data["y"], data["y"][:1]])[:nx]) if nx > 200 else np.ascontiguousarray(data["y"][:nx])
    spectra = GaussLineSpectra(x, y, noise_level=0.01)
    dev = jointstate.GaussJointState(spectra, nlive, sample.kernel_params, shelf_cap=4, fetch_rows=fetch_rows is True,
                                     via_backend=fetch_rows == "backend")
    host = jointstate.HostJointState(LaneScorer(spectra), nlive, ndata, sample.kernel_params)
    xs0 = sample.priortransform_batch(rng.uniform(size=(nlive, 3)))
    dev.init(xs0)
    host.init(xs0)
    assert np.array_equal(dev.live_matrix(), host.live_matrix())
    ndraws = _drive(dev, host, ndata, rng, iterations=12 if ndata <= 1000 else 5, exact=True)
    assert ndraws > 0
    dev.close()


def test_joint_state_against_the_cpu_oracle(oracle):
    from oracle_backend import OracleSpectra
    ndata, nlive = 300, 30
    rng = np.random.RandomState(5)
    data = gen.nothing(ndata)
    spectra = GaussLineSpectra(data["x"], data["y"], noise_level=0.01)
    dev = jointstate.GaussJointState(spectra, nlive, sample.kernel_params)
    host = jointstate.HostJointState(OracleSpectra(oracle, data["x"], data["y"]), nlive, ndata, sample.kernel_params)
    xs0 = sample.priortransform_batch(rng.uniform(size=(nlive, 3)))
    dev.init(xs0)
    host.init(xs0)
    assert _drive(dev, host, ndata, rng, iterations=8, exact=False) > 0
    dev.close()


def test_shelves_grow_past_their_first_capacity():
    """More accepted points waiting than the handle was created for: mdns_joint_reserve."""
    ndata, nlive = 64, 10
    rng = np.random.RandomState(3)
    data = gen.nothing(ndata)
    spectra = GaussLineSpectra(data["x"], data["y"], noise_level=0.01)
    dev = jointstate.GaussJointState(spectra, nlive, sample.kernel_params, shelf_cap=4)
    host = jointstate.HostJointState(LaneScorer(spectra), nlive, ndata, sample.kernel_params)
    xs0 = sample.priortransform_batch(rng.uniform(size=(nlive, 3)))
    dev.init(xs0)
    host.init(xs0)
    dev.prepare()
    host.prepare()
    lengths = []
    for k in range(40):
        cube = rng.uniform(size=(50, 3))
        cube[:, 0] *= 0.02
        xs = sample.priortransform_batch(cube)
        ia, La, ba, na = dev.draw(xs, None)
        ib, Lb, bb, _ = host.draw(xs[:na], None)
        assert ia == ib
        if ia >= 0:
            assert np.array_equal(ba, bb) and np.array_equal(La, Lb)
        if k in (9, 39):
            # the start of an iteration with shelves this long: nothing is purged (the live points did
            # not change), and the thresholds -- the (n+1)-th smallest of live + shelf, by quickselect
            # among the sixteen lanes of a data set -- come out as the commits left them
            ha, hn = dev.thresholds()
            lengths.append((int(hn.min()), int(hn.max())))
            a, b = dev.prepare(), host.prepare()
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] is None and b[2] is None
            ha2, hn2 = dev.thresholds()
            assert np.array_equal(hn2, hn) and np.array_equal(ha2, ha)
    ha, hn = dev.thresholds()
    hb, hm = host.thresholds()
    assert hn.max() > 4, "the test did not fill a shelf past the first capacity"
    assert np.array_equal(hn, hm) and np.array_equal(ha, hb)
    assert 0 < lengths[0][1] < 16 <= lengths[1][0], ("short and long shelves must be exercised", lengths)
    dev.close()


def test_long_shelves_purge_and_threshold():
    """Shelves longer than the 128 entries a data set's sixteen lanes hold in registers
    (k_joint_prepare): more than 128 accepted points waiting, then iterations -- every one replaces the worst
    live point by the head of the shelf, purges what no longer beats the new minimum (entries move
    down, in order) and selects the threshold among live + shelf -- against the numpy statement."""
    ndata, nlive = 70, 12
    rng = np.random.RandomState(5)
    data = gen.nothing(ndata)
    spectra = GaussLineSpectra(data["x"], data["y"], noise_level=0.01)
    dev = jointstate.GaussJointState(spectra, nlive, sample.kernel_params, shelf_cap=4)
    host = jointstate.HostJointState(LaneScorer(spectra), nlive, ndata, sample.kernel_params)
    xs0 = sample.priortransform_batch(rng.uniform(size=(nlive, 3)))
    dev.init(xs0)
    host.init(xs0)
    dev.prepare()
    host.prepare()
    for k in range(300):
        if k % 3 != 2:
            # ever fainter lines at one place: on spectra of pure noise every such draw beats the one
            # before until the amplitude reaches the noise, so the shelves of all data sets keep growing
            cube = np.column_stack([0.3 * 0.985 ** k * rng.uniform(0.9, 1.0, size=6), np.full(6, 0.5), np.full(6, 0.5)])
        else:
            # and lines anywhere, which some data sets take and others do not: shelves out of order
            cube = rng.uniform(size=(20, 3))
            cube[:, 0] *= 0.3 * 0.985 ** k
        xs = sample.priortransform_batch(cube)
        ia, La, ba, na = dev.draw(xs, None)
        ib, Lb, bb, _ = host.draw(xs[:na], None)
        assert ia == ib
    # the same (best) candidate again and again: equal likelihoods wait in the shelves (ties at the
    # threshold, which is then one of the copies); the thresholds see to it that the purge never
    # finds anything to drop -- what is kept and where is compared all the same
    cube = np.column_stack([np.full(3, 0.3 * 0.985 ** 300), np.full(3, 0.5), np.full(3, 0.5)])
    for k in range(20):
        xs = sample.priortransform_batch(cube)
        ia, La, ba, na = dev.draw(xs, None)
        ib, Lb, bb, _ = host.draw(xs[:na], None)
        assert ia == ib
    ha, hn = dev.thresholds()
    assert hn.max() > 128, ("the shelves did not grow past the registers", int(hn.max()))
    for it in range(int(hn.max())):
        if hn.min() <= 1:
            break
        dev.advance()
        host.advance()
        a, b = dev.prepare(), host.prepare()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        ha, hn = dev.thresholds()
        hb, hm = host.thresholds()
        assert np.array_equal(hn, hm), it
        assert np.array_equal(ha, hb), it
        assert np.array_equal(dev.live_matrix(), host.live_matrix())
    assert it > 40
    dev.close()


def test_commit_without_its_score_is_refused(hip):
    """Flags and trail of the accept pass belong to one score: a second commit, a commit that names
    another selection size, or a host-pointer commit with nothing scored fail with a message."""
    from massivedatans_amd import _lib
    import ctypes as C
    ndata, nlive = 130, 8
    rng = np.random.RandomState(2)
    data = gen.horns(ndata)
    spectra = GaussLineSpectra(data["x"], data["y"], noise_level=0.01)
    js = jointstate.GaussJointState(spectra, nlive, sample.kernel_params, fetch_rows=False)
    js.init(sample.priortransform_batch(rng.uniform(size=(nlive, 3))))
    js.prepare()
    accepted = C.c_int(0)
    assert hip.mdns_joint_commit(js._h, C.byref(accepted), None, None) != 0        # nothing scored yet
    assert b"score" in hip.mdns_last_error()
    js.draw(sample.priortransform_batch(rng.uniform(size=(5, 3))), None)             # score + commit
    assert hip.mdns_joint_commit_bits_dev(js._h, None, ndata) != 0                   # that score is used up
    assert b"score" in hip.mdns_last_error()
    js.draw(sample.priortransform_batch(rng.uniform(size=(5, 3))), None)             # and the state still works
    js.close()


def test_chunk_with_no_acceptable_candidate_changes_nothing():
    ndata, nlive = 200, 20
    rng = np.random.RandomState(11)
    data = gen.horns(ndata)
    spectra = GaussLineSpectra(data["x"], data["y"], noise_level=0.01)
    dev = jointstate.GaussJointState(spectra, nlive, sample.kernel_params)
    # live points: faint lines (good fits); candidates: the brightest, broadest lines the prior has
    cube = rng.uniform(size=(nlive, 3))
    cube[:, 0] *= 0.01
    dev.init(sample.priortransform_batch(cube))
    dev.prepare()
    before = dev.thresholds()
    bad = np.column_stack([np.full(90, 1.0), rng.uniform(size=90), np.full(90, 1.0)])
    idx, L, beats, n = dev.draw(sample.priortransform_batch(bad), None)
    assert idx == -1 and L is None and n == 90
    after = dev.thresholds()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    rows = np.array([3, 50, 199])
    idx, L, beats, n = dev.draw(sample.priortransform_batch(bad), rows)
    assert idx == -1
    dev.close()


@pytest.mark.parametrize("mode", ["1", "mfma", "mfma-direct", "mfma-lds", "default"])
def test_guarded_filter_decides_like_the_chain(hip, monkeypatch, mode):
    """Issue-bound chunks (256 candidates x 10 000 spectra) take the accept test as a guarded filter:
    one FMA per (candidate, channel, spectrum) on the expanded square, a rigorous error band around
    every threshold, and an exact re-score by the chain for whatever falls inside the band
    (k_gauss_cols_filter with vector FMAs; k_gauss_mfma_filter + k_exact_list on the
    matrix cores).  Its decisions must be the chain kernel's bit for bit
    -- also when a threshold is planted EXACTLY on a candidate's chain likelihood (L > thr is false)
    or one ulp below it (true): the cases only the resolve pass can get right."""
    from massivedatans_amd import _lib
    import subprocess, sys, os
    if mode == "default":
        # 1024 candidates x 10 000 spectra: the matrix-core filter is what the library picks by itself
        if os.environ.get("MDNS_K1_FILTER") is not None:
            pytest.skip("MDNS_K1_FILTER is set")
    elif os.environ.get("MDNS_K1_FILTER") != mode.split("-")[0] or os.environ.get("MDNS_K1_FILTER_FORM", "") != (mode.split("-") + [""])[1]:
        # the library reads the switches once per process: run this test in a child with the filter on
        # ("mfma": both operands straight from memory in tiles of 16 rows, the default form; "mfma-lds": staged through
        # LDS, round 3; "mfma-direct": straight from memory in the lane kernel's layouts)
        env = dict(os.environ, MDNS_K1_FILTER=mode.split("-")[0])
        env.pop("MDNS_K1_FILTER_FORM", None)
        if "-" in mode:
            env["MDNS_K1_FILTER_FORM"] = mode.split("-")[1]
        out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "gpu",
                              __file__ + "::test_guarded_filter_decides_like_the_chain[%s]" % mode], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
        return
    ndata, nlive, B = 10000, 100, (1024 if mode == "default" else 256)
    rng = np.random.RandomState(21)
    data = gen.horns(ndata)
    spectra = GaussLineSpectra(data["x"], data["y"], noise_level=0.01)
    cube = rng.uniform(size=(B, 3))
    xs = sample.priortransform_batch(cube)
    params = sample.kernel_params(xs)
    Lall = LaneScorer(spectra).loglike_batch(params)                  # the chain's values, [B, ndata]
    best = Lall.max(axis=0)
    who = Lall.argmax(axis=0)
    js = jointstate.GaussJointState(spectra, nlive, sample.kernel_params, fetch_rows=False)
    base = np.full((nlive, ndata), 0.0)
    base[0] = best + np.abs(best) * 1e-3                               # thresholds nobody beats
    planted = [(int(d), int(who[d])) for d in (17, 4242, 9999, 5000, 63, 64)]

    def run(live):
        _lib.check(hip.mdns_joint_set_live(js._h, _lib.ptr(np.ascontiguousarray(live))), "set_live")
        js.prepare()
        out = js.draw(xs, None)
        name = {"1": "k_gauss_cols_filter", "mfma-direct": "k_gauss_mfma_direct", "mfma-lds": "k_gauss_mfma_filter"}.get(mode, "k_gauss_gemm_filter")
        assert (hip.mdns_profile_kernel(0) or b"").decode().startswith(name), "the filter did not run"
        return out

    idx, _, beats, n = run(base)
    assert idx == -1 and n == B
    # thresholds exactly ON the best candidate's likelihood: still nobody (strict comparison)
    live = base.copy()
    for d, b in planted:
        live[0, d] = Lall[b, d]
    idx, _, beats, n = run(live)
    assert idx == -1
    # one ulp below: the first of those candidates is accepted, by exactly the planted data sets it tops
    live = base.copy()
    for d, b in planted:
        live[0, d] = np.nextafter(Lall[b, d], -np.inf)
    idx, _, beats, n = run(live)
    first = min(b for _, b in planted)
    assert idx == first
    want = np.zeros(ndata, dtype=bool)
    for d, b in planted:
        want[d] = Lall[first, d] > live[0, d]
    assert np.array_equal(beats, Lall[first] > live.min(axis=0)) and np.array_equal(beats, want) and beats.sum() >= 1
    # and an ordinary chunk: thresholds in the middle of the candidates' range
    live = base.copy()
    live[0] = np.percentile(Lall, 99.9, axis=0)
    idx, _, beats, n = run(live)
    ok = (Lall > live[0]).any(axis=1)
    assert idx == int(np.argmax(ok)) and np.array_equal(beats, Lall[idx] > live[0])
    js.close()
