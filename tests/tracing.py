"""Test helper: run a sampler under the integrator the way oracle/make_trace.py ran the
reference, keeping the same per-iteration record, and compare it with a golden trace."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_trace(case):
    with np.load(os.path.join(ROOT, "tests", "golden", "trace_%s.npz" % case)) as f:
        g = {k: f[k] for k in f.files}
    g["_name"] = case
    return g


class Recorder(object):
    """Forwards to the sampler, keeps what every next() returned and -- like the recorder of
    oracle/make_trace.py -- the live-point columns of every data set as they stood when the
    integrator's cut_down removed it (after the last cut_down the sampler's own matrices are
    [nlive, 0])."""

    def __init__(self, sampler):
        nlive, ndata = sampler.live_pointsp.shape
        self.__dict__.update(_s=sampler, Ls=[], us=[], ndraws_after=[], active=np.ones(ndata, dtype=bool),
                             term_p=np.full((nlive, ndata), -1, dtype=int), term_L=np.full((nlive, ndata), np.nan))

    def __getattr__(self, name):
        return getattr(self._s, name)

    def __setattr__(self, name, value):
        setattr(self._s, name, value)

    def cut_down(self, surviving):
        surviving = np.asarray(surviving, dtype=bool)
        leaving = np.flatnonzero(self.active)[~surviving]
        self.term_p[:, leaving] = np.asarray(self._s.live_pointsp)[:, ~surviving]
        self.term_L[:, leaving] = np.asarray(self._s.live_pointsL)[:, ~surviving]
        self.active[leaving] = False
        return self._s.cut_down(surviving)

    def __next__(self):
        u, x, L = next(self._s)
        self.Ls.append(np.array(L))
        self.us.append(np.array(u))
        self.ndraws_after.append(int(self._s.ndraws))
        return u, x, L

    next = __next__


def check_bookkeeping(g, sampler, rec, results):
    """Integer side of a run against the trace: must be exact whatever computed the likelihoods."""
    assert np.array_equal(np.array([len(L) for L in rec.Ls]), g["iter_nrunning"])
    assert np.array_equal(np.array(rec.ndraws_after), g["iter_ndraws"])
    assert sampler.ndraws == int(g["ndraws"])
    assert len(sampler.pointpile) == int(g["npoints"])
    want = g["final_live_pointsp"]
    assert want.shape == (int(g["nlive"]), int(g["ndata"])) and (want >= 0).all(), "vacuous fixture"
    assert np.array_equal(rec.term_p, want)
    assert len(results["weights"]) == int(g["nweights"])
    if "iter_u" in g:
        assert np.array_equal(np.concatenate(rec.us), g["iter_u"])       # accepted points: host arithmetic


def check_floats(g, rec, results, rtol):
    """Floating-point side: rtol = 0 demands the same bits (CPU oracle backends)."""
    def close(a, b):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        if rtol == 0:
            return np.array_equal(a, b)
        return a.shape == b.shape and bool(np.all(np.abs(a - b) <= rtol * np.abs(b)))
    if "iter_L" in g:
        assert close(np.concatenate(rec.Ls), g["iter_L"])
    assert close(rec.term_L, g["final_live_pointsL"])
    assert close(results["logZ"], g["logZ"])
    if rtol == 0:
        assert np.array_equal(results["information"], g["information"])
        assert np.array_equal(results["logZerr"], g["logZerr"])
    else:
        # (differences of nearly equal numbers: the information H and the error derived from it)
        assert np.allclose(results["information"], g["information"], rtol=max(rtol, 1e-6), atol=1e-9)
        assert np.allclose(results["logZerr"], g["logZerr"], rtol=max(rtol, 1e-6), atol=1e-9)
