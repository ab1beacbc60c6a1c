"""The floating-point state of the joint sampler behind one interface, twice.

The reference keeps the likelihoods of all live points (``live_pointsL[nlive, ndata]``,
multi_nested_sampler.py:111) and of the accepted points waiting on the shelves (:117) in
Python, and derives from them on every draw the thresholds ``Lmins_higher`` (:438-447), the
accept test ``any(L > Lmins)`` (hiermetriclearn.py:193) and the shelf fill (:482-485).  The
sampler of this package can hand all of that to a *joint state* object and keep only the
integer side (point ids, queues of ids, the data-set graph):

* :class:`GaussJointState` -- the state lives in HBM (``mdns_joint_*`` of include/mdns.h); a
  draw chunk is scored, decided and committed on the GPU and only the accepted candidate's
  index, likelihood row and fill bits come back.
* :class:`HostJointState` -- the same interface in numpy over any ``loglike_batch`` scorer.  It
  is the statement the device implementation is tested against (tests/), and what the CPU
  tests run the orchestration with.

Interface (``rows`` are ORIGINAL data-set indices, ascending; ``running`` likewise):

    init(xs)                     score the initial live points (physical parameter rows)
    set_running(running)         the data sets still being sampled (cut_down)
    prepare() -> Lmin, argmin, keep     start of an iteration; keep[r, e] = shelf entry e of the
                                        r-th running data set stays (bool matrix as wide as the
                                        longest shelf; None when nothing was dropped anywhere)
    draw(xs, rows) -> idx, Lrow, beats, nscored
                                 first acceptable candidate of the chunk (or -1), its
                                 likelihoods over ``rows``, which of them it beats; the state
                                 takes the point in; ``nscored`` candidates were looked at
    advance()                    end of an iteration
    live_matrix() -> L[nlive, nrunning]
"""
import ctypes as C

import numpy

from . import _host, _lib


class HostJointState(object):
    """numpy statement of the joint state over ``scorer.loglike_batch(params[B, 3], mask)``
    (``params`` rows are (A, mu, sig), ``mask`` a bool array over all data sets)."""

    #: candidates scored at once grow 1, 2, 4 ... up to this (a CPU scorer pays per candidate)
    MAX_CHUNK = 64

    def __init__(self, scorer, nlive, ndata, to_kernel_params, nparams=3):
        self.scorer = scorer
        self.nlive, self.ndata = int(nlive), int(ndata)
        self.to_kernel_params = to_kernel_params
        self.nparams = int(nparams)
        self.live = None                                  # [nlive, ndata], all data sets ever
        self.shelfL = [[] for _ in range(self.ndata)]
        self.higher = numpy.full(self.ndata, numpy.nan)
        self.running = numpy.arange(self.ndata)
        self.argmin = numpy.zeros(self.ndata, dtype=int)
        self.nevals_scored = 0
        self.ncalls = 0

    def init(self, xs, jitter=None):
        """``jitter`` [nlive, ndata]: noise added to the likelihoods (musefuse.py:535)."""
        self.live = numpy.array(self.scorer.loglike_batch(self.to_kernel_params(xs), numpy.ones(self.ndata, dtype=bool)))
        if jitter is not None:
            self.live = self.live + jitter
        self.nevals_scored += self.live.size
        self.ncalls += 1
        assert self.live.shape == (self.nlive, self.ndata)

    def set_running(self, running):
        self.running = numpy.asarray(running, dtype=int)

    def _threshold(self, d):
        n = len(self.shelfL[d])
        merged = numpy.concatenate((self.live[:, d], self.shelfL[d]))
        return numpy.partition(merged, n)[n]              # find_nsmallest, multi_nested_sampler.py:44-47

    def prepare(self):
        run = self.running
        cols = self.live[:, run]
        Lmin = cols.min(axis=0)
        arg = cols.argmin(axis=0)
        self.argmin[run] = arg
        width = max([len(self.shelfL[d]) for d in run] + [0])
        keep = numpy.zeros((len(run), width), dtype=bool)
        dropped = False
        for r, d in enumerate(run):
            k = [L > Lmin[r] for L in self.shelfL[d]]
            dropped = dropped or not all(k)
            keep[r, :len(k)] = k
            self.shelfL[d] = [L for L, kk in zip(self.shelfL[d], k) if kk]
            self.higher[d] = self._threshold(d) if self.shelfL[d] else Lmin[r]
        return Lmin, arg, (keep if dropped else None)

    def chunk_size(self, offered, M, hint=None):
        return offered

    def took(self, rows, beats):
        """(a native constrainer's draw ended in this state's ``draw_params``: nothing to add)"""

    def draw(self, xs, rows):
        return self.draw_params(self.to_kernel_params(xs), rows)

    def draw_params(self, params, rows, jitter=None):
        """``draw`` for candidates given as kernel parameter rows; ``jitter`` [B, M] is added to
        their likelihoods before anything is compared or kept."""
        rows = numpy.arange(self.ndata) if rows is None else numpy.asarray(rows, dtype=int)
        mask = numpy.zeros(self.ndata, dtype=bool)
        mask[rows] = True
        thr = self.higher[rows]
        pos, chunk = 0, 1
        while pos < len(params):
            Ls = self.scorer.loglike_batch(params[pos:pos + chunk], mask)
            if jitter is not None:
                Ls = Ls + jitter[pos:pos + len(Ls)]
            self.nevals_scored += Ls.size
            self.ncalls += 1
            ok = (Ls > thr).any(axis=1)
            if ok.any():
                i = int(numpy.argmax(ok))
                Lrow = Ls[i]
                beats = Lrow > thr
                for d, L in zip(rows[beats], Lrow[beats]):
                    self.shelfL[d].append(L)
                    self.higher[d] = self._threshold(d)
                return pos + i, Lrow, beats, pos + i + 1
            pos += len(Ls)
            chunk = min(self.MAX_CHUNK, 2 * chunk)
        return -1, None, None, len(params)

    # the two halves of draw(), for a host that reduces the accept flags over several states
    # (parallel.ShardedJointState) in between
    def score(self, xs, rows):
        """Accept flag of every candidate of the chunk against the selected data sets."""
        return self.score_params(self.to_kernel_params(xs) if len(xs) else numpy.zeros((0, self.nparams)), rows)

    def score_params(self, params, rows, jitter=None):
        """``score`` for candidates given as kernel parameter rows; ``jitter`` [B, M] is added to the
        likelihoods before anything is compared or kept (musefuse.py:535)."""
        xs = params
        rows = numpy.arange(self.ndata) if rows is None else numpy.asarray(rows, dtype=int)
        self._scored_rows = rows
        if len(rows) == 0 or len(xs) == 0:
            self._scored_L = numpy.zeros((len(xs), 0))
            return numpy.zeros(len(xs), dtype=numpy.int32)
        mask = numpy.zeros(self.ndata, dtype=bool)
        mask[rows] = True
        self._scored_L = self.scorer.loglike_batch(params, mask)
        if jitter is not None:
            self._scored_L = self._scored_L + numpy.asarray(jitter)[:len(self._scored_L)]
        self.nevals_scored += self._scored_L.size
        self.ncalls += 1
        return (self._scored_L > self.higher[rows]).any(axis=1).astype(numpy.int32)

    def commit(self, idx):
        """Candidate ``idx`` of the scored chunk is the accepted point: its likelihoods over the
        scored selection, which data sets it beats; those take it in."""
        rows = self._scored_rows
        Lrow = self._scored_L[idx]
        beats = Lrow > self.higher[rows]
        for d, L in zip(rows[beats], Lrow[beats]):
            self.shelfL[d].append(L)
            self.higher[d] = self._threshold(d)
        return Lrow, beats

    def advance(self):
        for d in self.running:
            self.live[self.argmin[d], d] = self.shelfL[d].pop(0)

    def live_matrix(self):
        return self.live[:, self.running]

    def thresholds(self):
        return self.higher.copy(), numpy.array([len(s) for s in self.shelfL])


class GaussJointState(object):
    """The joint state of the Gaussian-line problem on the GPU, bound to a
    :class:`massivedatans_amd.like.GaussLineSpectra` (include/mdns.h Part 2b)."""

    #: (candidate, spectrum) pairs scored per chunk at most: ~50 us of GPU time
    EVAL_BUDGET = 2560000
    MIN_CHUNK = 32

    nparams = 3

    def __init__(self, spectra, nlive, to_kernel_params, shelf_cap=64, fetch_rows=True, via_backend=False):
        #: copy the accepted candidate's likelihood row to the host with every draw (the sampler
        #: itself needs only the index and the fill bits: sample.py turns this off)
        self.fetch_rows = fetch_rows
        #: make ``draw`` go through the entry points a native constrainer calls
        #: (mdns_backend_draw_begin / mdns_backend_draw_chunk; no likelihood row) -- tests
        self.via_backend = via_backend
        self._nscored = C.c_int(0)
        self._lib = _lib.require_device()
        self.spectra = spectra                             # keeps the spectra handle alive
        self.nlive, self.ndata = int(nlive), int(spectra.ndata)
        self.noise_level = float(getattr(spectra, "noise_level", 0.0))
        self.to_kernel_params = to_kernel_params
        self._h = self._lib.mdns_joint_create(spectra.handle, self.nlive, int(shelf_cap))
        if not self._h:
            raise _lib.MdnsError("mdns_joint_create failed: " + _lib.last_error())
        self.running = numpy.arange(self.ndata, dtype=numpy.int32)
        self.shelf_n = numpy.zeros(self.ndata, dtype=numpy.int64)     # mirror of the device's shelf sizes
        self.cap = self._lib.mdns_joint_shelf_cap(self._h)
        self._Lrow = numpy.empty(self.ndata)
        self._bits = numpy.zeros((self.ndata + 63) // 64, dtype=numpy.uint64)
        self._accepted = C.c_int(-1)
        self.nevals_scored = 0
        self.ncalls = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mdns_joint_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise _lib.MdnsError("%s failed: %s" % (what, _lib.last_error()))

    def init(self, xs):
        params = _lib.as_f64(self.to_kernel_params(xs))
        if params.shape != (self.nlive, 3):
            raise ValueError("initial points must be [nlive, 3]")
        self._check(self._lib.mdns_joint_init_gauss(self._h, _lib.ptr(params), self.noise_level), "mdns_joint_init_gauss")
        self.shelf_n[:] = 0
        self.nevals_scored += self.nlive * self.ndata
        self.ncalls += 1

    def set_running(self, running):
        self.running = numpy.ascontiguousarray(running, dtype=numpy.int32)
        self._check(self._lib.mdns_joint_set_running(self._h, _lib.ptr(self.running), len(self.running)),
                    "mdns_joint_set_running")

    def prepare(self):
        n = len(self.running)
        kw = self._lib.mdns_joint_keep_words(self._h)
        Lmin = numpy.empty(n)
        arg = numpy.empty(n, dtype=numpy.int32)
        keepw = numpy.empty((n, kw), dtype=numpy.uint64)
        self._check(self._lib.mdns_joint_prepare(self._h, _lib.ptr(Lmin), _lib.ptr(arg), _lib.ptr(keepw)),
                    "mdns_joint_prepare")
        counts = self.shelf_n[self.running]
        keep = None
        if counts.any():
            # kept entries per data set = set bits; anything dropped shows as a smaller count
            kept = numpy.zeros(n, dtype=numpy.int64)
            for w in range(kw):
                kept += _popcount(keepw[:, w])
            if (kept != counts).any():
                width = int(counts.max())
                keep = numpy.zeros((n, width), dtype=bool)
                for e in range(width):
                    keep[:, e] = (keepw[:, e // 64] >> numpy.uint64(e % 64)) & numpy.uint64(1)
                self.shelf_n[self.running] = kept
        return Lmin, arg.astype(int), keep

    def chunk_size(self, offered, M, hint=None):
        """How many of the offered candidates one launch scores: four times the tries the last
        draw needed (``hint``), within a budget of (candidate, spectrum) pairs."""
        budget = max(self.MIN_CHUNK, self.EVAL_BUDGET // max(1, M))
        if hint is not None:
            budget = min(budget, max(self.MIN_CHUNK, 4 * int(hint)))
        return int(min(offered, budget, _lib.JOINT_MAX_BATCH))

    def took(self, rows, beats):
        """A draw made through the native constrainer (mdns_backend_draw_chunk on this state's
        handle) put its point on the shelves of ``rows[beats]``: keep the mirror of the sizes."""
        if rows is None:
            self.shelf_n[beats] += 1
        else:
            self.shelf_n[rows[beats]] += 1

    def draw(self, xs, rows):
        B = min(len(xs), _lib.JOINT_MAX_BATCH)
        return self.draw_params(self.to_kernel_params(xs[:B]), rows)

    def draw_params(self, params, rows, jitter=None):
        """``draw`` for candidates given as kernel parameter rows (A, mu, sig)."""
        M = self.ndata if rows is None else len(rows)
        if jitter is not None:
            if not self.via_backend:
                raise ValueError("likelihood jitter goes through the backend entry points")
            jitter = _lib.as_f64(jitter)
        B = min(len(params), _lib.JOINT_MAX_BATCH)
        params = _lib.as_f64(params[:B])
        if rows is not None:
            rows = numpy.ascontiguousarray(rows, dtype=numpy.int32)
        if self.via_backend:
            if B == 0 or M == 0:
                return -1, None, None, B
            self._check(self._lib.mdns_backend_draw_begin(self._h, _lib.ptr(rows) if rows is not None else None, M),
                        "mdns_backend_draw_begin")
            self._check(self._lib.mdns_backend_draw_chunk(self._h, _lib.ptr(params), B,
                                                          _lib.ptr(jitter) if jitter is not None else None,
                                                          C.addressof(self._accepted), _lib.ptr(self._bits), C.addressof(self._nscored)),
                        "mdns_backend_draw_chunk")
            self.ncalls += 1
            self.nevals_scored += B * M
            idx = self._accepted.value
            if idx < 0:
                return -1, None, None, B
            beats = numpy.unpackbits(self._bits[:(M + 63) // 64].view(numpy.uint8), bitorder='little')[:M].astype(bool)
            self.took(rows, beats)
            return idx, None, beats, B
        if rows is not None:
            nmax = int(self.shelf_n[rows].max()) if M else 0
        else:
            nmax = int(self.shelf_n.max())
        if nmax + 1 > self.cap:
            self._check(self._lib.mdns_joint_reserve(self._h, nmax + 1), "mdns_joint_reserve")
            self.cap = self._lib.mdns_joint_shelf_cap(self._h)
        self._check(self._lib.mdns_joint_draw_gauss(
            self._h, _lib.ptr(params), B, self.noise_level, _lib.ptr(rows) if rows is not None else None, M,
            C.byref(self._accepted), _lib.ptr(self._Lrow) if self.fetch_rows else None, _lib.ptr(self._bits)),
            "mdns_joint_draw_gauss")
        self.ncalls += 1
        idx = self._accepted.value
        if idx < 0:
            self.nevals_scored += B * M
            return -1, None, None, B
        self.nevals_scored += B * M
        beats = numpy.unpackbits(self._bits[:(M + 63) // 64].view(numpy.uint8), bitorder='little')[:M].astype(bool)
        if rows is None:
            self.shelf_n[beats] += 1
        else:
            self.shelf_n[rows[beats]] += 1
        return idx, (self._Lrow[:M].copy() if self.fetch_rows else None), beats, B

    # the two halves of a chunk through the entry points a native constrainer uses (include/mdns.h:
    # mdns_backend_draw_score / mdns_backend_draw_commit): what parallel.ShardedJointState puts the MAX
    # all-reduce of the candidates' votes between.  Both kinds of state (Gaussian line, MUSE-style).
    def score_backend(self, params, rows, jitter=None):
        """Scores the chunk against the selected data sets ``rows`` (ORIGINAL indices of this state's
        spectra, ascending; None: all); one 0 / 1 vote per candidate stays on the device
        (:meth:`votes_address`)."""
        B = len(params)
        if B > _lib.JOINT_MAX_BATCH:
            raise ValueError("at most %d candidates per chunk" % _lib.JOINT_MAX_BATCH)
        M = self.ndata if rows is None else len(rows)
        params = _lib.as_f64(params) if B else numpy.zeros((0, self.nparams))
        if rows is not None:
            rows = numpy.ascontiguousarray(rows, dtype=numpy.int32)
        if jitter is not None:
            jitter = _lib.as_f64(jitter)
            if jitter.shape != (B, M):
                raise ValueError("jitter must be [B, M]")
        self._check(self._lib.mdns_backend_draw_begin(self._h, _lib.ptr(rows) if rows is not None and M else None, M),
                    "mdns_backend_draw_begin")
        self._check(self._lib.mdns_backend_draw_score(self._h, _lib.ptr(params) if B else None, B,
                                                      _lib.ptr(jitter) if jitter is not None and B * M else None),
                    "mdns_backend_draw_score")
        self._half = (rows, M, B)
        self.nevals_scored += B * M
        self.ncalls += 1

    def votes_address(self):
        return self._lib.mdns_joint_votes_dev(self._h)

    def votes(self):
        rows, M, B = self._half
        out = numpy.zeros(B, dtype=numpy.int32)
        if B:
            self._check(self._lib.mdns_d2h(_lib.ptr(out), self.votes_address(), out.nbytes), "mdns_d2h")
        return out

    def set_votes(self, votes):
        votes = numpy.ascontiguousarray(votes, dtype=numpy.int32)
        if len(votes):
            self._check(self._lib.mdns_h2d(self.votes_address(), _lib.ptr(votes), votes.nbytes), "mdns_h2d")

    def commit_backend(self):
        """The first candidate that has a vote now is the accepted point: (its index or -1, which of the
        selected data sets of THIS state it beats)."""
        rows, M, B = self._half
        self._check(self._lib.mdns_backend_draw_commit(self._h, C.addressof(self._accepted), _lib.ptr(self._bits)),
                    "mdns_backend_draw_commit")
        idx = self._accepted.value
        if idx < 0 or M == 0:
            return idx, numpy.zeros(M, dtype=bool)
        beats = numpy.unpackbits(self._bits[:(M + 63) // 64].view(numpy.uint8), bitorder='little')[:M].astype(bool)
        self.took(rows, beats)
        return idx, beats

    # the two halves of draw() (include/mdns.h: mdns_joint_score / mdns_joint_commit)
    def _reserve_for(self, rows):
        nmax = int(self.shelf_n.max()) if rows is None else (int(self.shelf_n[rows].max()) if len(rows) else 0)
        if nmax + 1 > self.cap:
            self._check(self._lib.mdns_joint_reserve(self._h, nmax + 1), "mdns_joint_reserve")
            self.cap = self._lib.mdns_joint_shelf_cap(self._h)

    def score(self, xs, rows):
        """Scores the chunk; the accept flags stay on the device (``flags`` / ``set_flags`` /
        ``flags_address`` reach them)."""
        return self.score_params(self.to_kernel_params(xs) if len(xs) else numpy.zeros((0, 3)), rows)

    def score_params(self, params, rows):
        """``score`` for candidates given as kernel parameter rows (A, mu, sig)."""
        xs = params
        B = len(xs)
        if B > _lib.JOINT_MAX_BATCH:
            raise ValueError("at most %d candidates per chunk" % _lib.JOINT_MAX_BATCH)
        if rows is not None:
            rows = numpy.ascontiguousarray(rows, dtype=numpy.int32)
        M = self.ndata if rows is None else len(rows)
        self._reserve_for(rows)
        params = _lib.as_f64(params) if B else numpy.zeros((0, 3))
        self._scored_rows, self._scored_B = rows, B
        self._check(self._lib.mdns_joint_score(self._h, _lib.ptr(params), B, self.noise_level,
                                               _lib.ptr(rows) if rows is not None and M else None, M), "mdns_joint_score")
        self.nevals_scored += B * M
        self.ncalls += 1

    def flags_address(self):
        return self._lib.mdns_joint_flags_dev(self._h)

    def flags(self):
        out = numpy.zeros(self._scored_B, dtype=numpy.int32)
        if self._scored_B:
            self._check(self._lib.mdns_d2h(_lib.ptr(out), self.flags_address(), out.nbytes), "mdns_d2h")
        return out

    def set_flags(self, flags):
        flags = numpy.ascontiguousarray(flags, dtype=numpy.int32)
        if len(flags):
            self._check(self._lib.mdns_h2d(self.flags_address(), _lib.ptr(flags), flags.nbytes), "mdns_h2d")

    def commit(self, idx=None):
        """The first flagged candidate (``idx`` is only checked against it) is the accepted point."""
        rows = self._scored_rows
        M = self.ndata if rows is None else len(rows)
        self._check(self._lib.mdns_joint_commit(self._h, C.byref(self._accepted),
                                                _lib.ptr(self._Lrow) if self.fetch_rows else None, _lib.ptr(self._bits)),
                    "mdns_joint_commit")
        got = self._accepted.value
        if M == 0:
            return (numpy.zeros(0) if self.fetch_rows else None), numpy.zeros(0, dtype=bool)
        if got < 0 or (idx is not None and got != idx):
            raise _lib.MdnsError("mdns_joint_commit accepted candidate %d, expected %s" % (got, idx))
        beats = numpy.unpackbits(self._bits[:(M + 63) // 64].view(numpy.uint8), bitorder='little')[:M].astype(bool)
        if rows is None:
            self.shelf_n[beats] += 1
        else:
            self.shelf_n[rows[beats]] += 1
        return (self._Lrow[:M].copy() if self.fetch_rows else None), beats

    def advance(self):
        self._check(self._lib.mdns_joint_advance(self._h), "mdns_joint_advance")
        self.shelf_n[self.running] -= 1

    def live_matrix(self):
        full = numpy.empty((self.nlive, self.ndata))
        self._check(self._lib.mdns_joint_get_live(self._h, _lib.ptr(full)), "mdns_joint_get_live")
        if len(self.running) == self.ndata:
            return full
        return full[:, self.running]

    def thresholds(self):
        higher = numpy.empty(self.ndata)
        n = numpy.empty(self.ndata, dtype=numpy.int32)
        self._check(self._lib.mdns_joint_get_thresholds(self._h, _lib.ptr(higher), _lib.ptr(n)), "mdns_joint_get_thresholds")
        return higher, n.astype(int)


class MuseJointState(GaussJointState):
    """The joint state of the MUSE-style problem on the GPU: spectra with per-pixel variances
    (:class:`massivedatans_amd.like.MuseSpectra`), the three-line template evaluated on the device
    from 5 parameters, the scale-marginalised likelihood of cmuselike.c:45-64.  Draw chunks go
    through the entry points a native constrainer calls (``mdns_backend_draw_*``)."""

    nparams = 5

    def __init__(self, spectra, nlive, shelf_cap=64):
        super(MuseJointState, self).__init__(spectra, nlive, lambda xs: xs, shelf_cap=shelf_cap, fetch_rows=False, via_backend=True)

    def init(self, xs, jitter=None):
        params = _lib.as_f64(xs)
        if params.shape != (self.nlive, 5):
            raise ValueError("initial points must be [nlive, 5]")
        if jitter is not None:
            jitter = _lib.as_f64(jitter)
            if jitter.shape != (self.nlive, self.ndata):
                raise ValueError("jitter must be [nlive, ndata]")
        self._check(self._lib.mdns_joint_init_muse3(self._h, _lib.ptr(params), _lib.ptr(jitter) if jitter is not None else None),
                    "mdns_joint_init_muse3")
        self.shelf_n[:] = 0
        self.nevals_scored += self.nlive * self.ndata
        self.ncalls += 1

    def chunk_size(self, offered, M, hint=None):
        return int(self._lib.mdns_backend_chunk_size(self._h, int(offered), int(M), int(hint or 1)))


_POP8 = numpy.array([bin(i).count("1") for i in range(256)], dtype=numpy.int64)


def _popcount(words):
    """Set bits of every uint64 of ``words``."""
    if hasattr(numpy, "bitwise_count"):
        return numpy.bitwise_count(numpy.ascontiguousarray(words)).astype(numpy.int64)
    return _POP8[numpy.ascontiguousarray(words).view(numpy.uint8).reshape(len(words), 8)].sum(axis=1)


__all__ = ['HostJointState', 'GaussJointState', 'MuseJointState']
