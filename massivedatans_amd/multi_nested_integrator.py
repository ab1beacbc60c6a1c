"""Nested-sampling evidence integration, vectorised over the data sets.

Host-side mirror of the reference's ``multi_nested_integrator.py:26-175``: same call
(``multi_nested_integrator(multi_sampler, tolerance, max_samples, min_samples)``), same result
dictionary (``logZ``, ``logZerr``, ``weights``, ``information``, ``niterations``) and the same
floating-point expressions in the same order, so that identical sampler output gives
bit-identical evidences.  The progress bar of the reference is not reproduced (it has no
effect on the results).
"""
import logging

import numpy
from numpy import exp, log, logaddexp

log_ = logging.getLogger("massivedatans_amd")


def integrate_remainder(sampler, logwidth, logVolremaining, logZ, H, globalLmax):
    """Contribution of the current live points, which all share the shell width ``logwidth``
    (multi_nested_integrator.py:26-59).  Returns (remainderZ, remainderZerr, totalZ, totalZerr,
    totalZerr) per running data set."""
    remainder = list(sampler.remainder())
    logV = logwidth
    L0 = globalLmax
    Ls = numpy.exp([Li - L0 for _, _, Li in remainder])
    LsMax = Ls.copy()
    LsMax[-1] = numpy.exp(globalLmax - L0)
    Lmax = LsMax[1:].sum(axis=0) + LsMax[-1]
    Lmin = Ls[:-1].sum(axis=0) + Ls[0]
    logLmid = log(Ls.sum(axis=0)) + L0
    logZmid = logaddexp(logZ, logV + logLmid)
    logZup = logaddexp(logZ, logV + log(Lmax) + L0)
    logZlo = logaddexp(logZ, logV + log(Lmin) + L0)
    logZerr = logZup - logZlo
    assert numpy.isfinite(H).all()
    assert numpy.isfinite(logZerr).all(), logZerr

    for _, _, Li in remainder:
        wi = logwidth + Li
        logZnew = logaddexp(logZ, wi)
        H = exp(wi - logZnew) * Li + exp(logZ - logZnew) * (H + logZ) - logZnew
        H[H < 0] = 0
        logZ = logZnew

    total_err = logZerr + (H / sampler.nlive_points) ** 0.5
    return logV + logLmid, logZerr, logZmid, total_err, total_err


def multi_nested_integrator(multi_sampler, tolerance=0.01, max_samples=None, min_samples=0,
                            need_robust_remainder_error=True):
    """Run ``multi_sampler`` until, for every data set, the evidence uncertainty (shell
    statistics + live-point remainder) is below ``tolerance`` (checked every 50 iterations,
    multi_nested_integrator.py:136), dropping finished data sets from the sampler as it goes."""
    sampler = multi_sampler
    nlive = sampler.nlive_points
    ndata = multi_sampler.ndata
    logVolremaining = 0
    logwidth = log(1 - exp(-1. / nlive))
    weights = []

    i = 0
    running = numpy.ones(ndata, dtype=bool)
    last_remainderZ = numpy.zeros(ndata)
    last_remainderZerr = numpy.zeros(ndata)
    logZerr = numpy.zeros(ndata)
    ui, xi, Li = next(sampler)
    wi = logwidth + Li
    logZ = wi
    H = Li - logZ
    ndim = ui.shape[1]
    # live points of each data set at its termination, in order of increasing likelihood
    tail_u = numpy.zeros((nlive, ndata, ndim))
    tail_x = numpy.zeros((nlive, ndata, ndim))
    tail_L = numpy.zeros((nlive, ndata))
    tail_w = numpy.zeros(ndata)
    while True:
        i = i + 1
        logwidth = log(1 - exp(-1. / nlive)) + logVolremaining
        logVolremaining -= 1. / nlive

        # one weighted sample per data set; finished data sets get zero weight
        Lifull = numpy.full(ndata, -numpy.inf)
        Lifull[running] = Li
        uifull = numpy.zeros((ndata, ui.shape[1]))
        uifull[running, :] = ui
        xifull = numpy.zeros((ndata, ui.shape[1]))
        xifull[running, :] = xi
        weights.append([uifull, xifull, Lifull, numpy.where(running, logwidth, -numpy.inf), running])

        logZerr[running] = (H[running] / nlive) ** 0.5

        if i > min_samples and i % 50 == 1 or (max_samples and i > max_samples):
            remainderZ, remainderZerr, totalZ, totalZerr, _ = integrate_remainder(
                sampler, logwidth, logVolremaining, logZ[running], H[running], sampler.Lmax)
            last_remainderZ[running] = remainderZ
            last_remainderZerr[running] = remainderZerr
            terminating = totalZerr < tolerance
            if max_samples and i > max_samples:
                terminating[:] = True
            if terminating.any():
                log_.debug('iteration %d: terminating %d data sets', i, terminating.sum())
                for j, k in enumerate(numpy.where(running)[0]):
                    if terminating[j]:
                        tail_u[:, k], tail_x[:, k], tail_L[:, k] = sampler.remainder_arrays(j)
                        tail_w[k] = logwidth
                sampler.cut_down(~terminating)
                running[running] = ~terminating
            if not running.any():
                break
        ui, xi, Li = next(sampler)
        wi = logwidth + Li
        logZnew = logaddexp(logZ[running], wi)
        H[running] = exp(wi - logZnew) * Li + exp(logZ[running] - logZnew) * (H[running] + logZ[running]) - logZnew
        logZ[running] = logZnew

    # the live points at termination complete the posterior sample (not needed for logZ)
    all_tails = numpy.ones(ndata, dtype=bool)
    for k in range(nlive):
        weights.append([tail_u[k], tail_x[k], tail_L[k], tail_w.copy(), all_tails])
    logZerr = logZerr + last_remainderZerr
    logZ = logaddexp(logZ, last_remainderZ)

    # the reference returns its loop variable after the tail loop, i.e. nlive-1
    # (SURVEY.md appendix A#12); sample.py reports len(weights) instead.  We keep that quirk
    # under the reference's key and add the true count.
    return dict(logZ=logZ, logZerr=logZerr, weights=weights, information=H,
                niterations=nlive - 1, nsamples=i)


__all__ = ['multi_nested_integrator', 'integrate_remainder']
