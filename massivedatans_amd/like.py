"""Host side of the likelihood hot path: spectra resident in HBM, candidates scored in batches.

Mirrors the problem-definition surface of the reference driver scripts:

* :class:`GaussLineSpectra` -- ``multi_loglikelihood(params, data_mask)`` of sample.py:101-108
  (one Gaussian emission line; clike.c).
* :class:`MuseSpectra` -- ``multi_loglikelihood_clike`` of musefuse.py:520-535 (scale-marginalised
  chi^2 against a template; cmuselike.c), without the RNG jitter (callers add it, see
  ``jitter``).

Both keep the spectra on the device as ``[n_datasets, n_channels]`` rows and add
``loglike_batch`` (B candidates per pass), which the reference does not have.
"""
import ctypes as C

import numpy as np

from . import _lib


def _rows_from_mask(data_mask, ndata):
    """Ascending indices of the selected spectra (None = all of them)."""
    if data_mask is None:
        return None, ndata
    m = np.asarray(data_mask)
    if m.dtype == np.bool_:
        if m.shape != (ndata,):
            raise ValueError("data_mask has shape %s, expected (%d,)" % (m.shape, ndata))
        if m.all():
            return None, ndata
        rows = np.flatnonzero(m).astype(np.int32)
    else:
        rows = np.ascontiguousarray(m, dtype=np.int32)
    return rows, len(rows)


class _Spectra(object):
    def __init__(self, x, y, v=None, layout="channel_major"):
        self._lib = _lib.require_device()
        y = _lib.as_f64(y)
        if y.ndim != 2:
            raise ValueError("y must be 2-D")
        if layout == "channel_major":        # reference layout [nx, ndata] (sample.py:31)
            nx, ndata = y.shape
            code = 0
        elif layout == "dataset_major":
            ndata, nx = y.shape
            code = 1
        else:
            raise ValueError(layout)
        self.ndata, self.nx = int(ndata), int(nx)
        xp = None
        if x is not None:
            x = _lib.as_f64(x)
            if x.shape != (nx,):
                raise ValueError("x has shape %s, expected (%d,)" % (x.shape, nx))
            xp = _lib.ptr(x)
        vp = None
        if v is not None:
            v = _lib.as_f64(v)
            if v.shape != y.shape:
                raise ValueError("v and y differ in shape")
            vp = _lib.ptr(v)
        self._h = self._lib.mdns_spectra_create(xp, _lib.ptr(y), vp, self.ndata, self.nx, code)
        if not self._h:
            raise _lib.MdnsError("mdns_spectra_create failed: " + _lib.last_error())

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mdns_spectra_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _batch(self, fn, name, params, data_mask, *extra):
        params = _lib.as_f64(params)
        B = params.shape[0]
        rows, M = _rows_from_mask(data_mask, self.ndata)
        out = np.empty((B, M))
        if B and M:
            args = [self._h, _lib.ptr(params), B] + list(extra) + \
                   [_lib.ptr(rows) if rows is not None else None, M, _lib.ptr(out)]
            _lib.check(fn(*args), name)
        return out


class GaussLineSpectra(_Spectra):
    """The toy problem of sample.py: ``ypred = A exp(-0.5 ((mu - x)/sig)^2)``,
    ``L_i = -0.5 sum_j ((ypred_j - y_ji)/noise_level)^2``."""

    def __init__(self, x, y, noise_level=0.01, layout="channel_major"):
        super(GaussLineSpectra, self).__init__(x, y, None, layout)
        self.noise_level = float(noise_level)

    def loglike_batch(self, params, data_mask=None):
        """``params[B, 3]`` = (A, mu, sig) with sig linear -> ``L[B, mask.sum()]``."""
        params = np.atleast_2d(_lib.as_f64(params))
        if params.shape[1] != 3:
            raise ValueError("params must be [B, 3] = (A, mu, sig)")
        return self._batch(self._lib.mdns_gauss_loglike_batch, "mdns_gauss_loglike_batch",
                           params, data_mask, self.noise_level)

    def multi_loglikelihood(self, params, data_mask):
        """Same call as sample.py:101-108: ``params = (A, mu, log10 sig)`` after
        priortransform; returns the likelihood vector of the masked data sets."""
        A, mu, log_sig = params
        sig = 10 ** log_sig
        return self.loglike_batch(np.array([[A, mu, sig]]), data_mask)[0]


class MuseSpectra(_Spectra):
    """Spectra with per-pixel variances scored against templates (cmuselike.c:45-64)."""

    def __init__(self, x, y, v, layout="channel_major"):
        super(MuseSpectra, self).__init__(x, y, v, layout)

    def loglike_batch(self, ypred, data_mask=None):
        """``ypred[B, nx]`` precomputed templates -> ``L[B, mask.sum()]`` (= -0.5 chi)."""
        ypred = np.atleast_2d(_lib.as_f64(ypred))
        if ypred.shape[1] != self.nx:
            raise ValueError("templates must be [B, %d]" % self.nx)
        return self._batch(self._lib.mdns_muse_loglike_batch, "mdns_muse_loglike_batch",
                           ypred, data_mask)

    def loglike_batch_lines(self, params, data_mask=None):
        """``params[B, 5]`` of the config-C5 three-line template, evaluated on the device."""
        params = np.atleast_2d(_lib.as_f64(params))
        if params.shape[1] != 5:
            raise ValueError("params must be [B, 5]")
        return self._batch(self._lib.mdns_muse3_loglike_batch, "mdns_muse3_loglike_batch",
                           params, data_mask)

    def multi_loglikelihood(self, ypred, data_mask, jitter=None):
        """musefuse.py:534-535 given the template: masked likelihoods; ``jitter`` (e.g.
        ``numpy.random.normal``) reproduces the reference's N(0, 1e-5) tie-breaker and its RNG
        consumption."""
        if not np.any(ypred):
            # "give low probability to solutions with no stars" (musefuse.py:527-529): no kernel
            # call and -- unlike the regular path -- no random numbers drawn
            return np.ones(_rows_from_mask(data_mask, self.ndata)[1]) * -1e100
        L = self.loglike_batch(ypred, data_mask)[0]
        if jitter is not None:
            L = L + jitter(0, 1e-5, size=len(L))
        return L
