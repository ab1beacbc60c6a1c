"""Nested-sampling evidence integration, vectorised over the data sets.

Host-side mirror of the reference's ``multi_nested_integrator.py:26-175``: same call
(``multi_nested_integrator(multi_sampler, tolerance, max_samples, min_samples)``), same result
dictionary (``logZ``, ``logZerr``, ``weights``, ``information``, ``niterations``) and the same
floating-point expressions evaluated in the same order, so that identical sampler output gives
bit-identical evidences (tests/test_orchestration.py).  The progress bar of the reference is not
reproduced (it has no effect on the results).

Per data set the integrator keeps the running evidence ``logZ`` and information ``H``; each
iteration contributes one shell of prior volume ``exp(-i/nlive) (1 - exp(-1/nlive))`` weighted
with the likelihood of the point that died, and every 50 iterations the live points are
integrated as a remainder to decide which data sets are done.
"""
import logging

import numpy
from numpy import exp, log, logaddexp

log_ = logging.getLogger("massivedatans_amd")


def _absorb(logZ, H, shell_logwidth, Li):
    """One weighted sample into (logZ, H): logZ' = logaddexp(logZ, w), and the information
    update of multi_nested_integrator.py:46,160 (same association of terms)."""
    wi = shell_logwidth + Li
    grown = logaddexp(logZ, wi)
    H = exp(wi - grown) * Li + exp(logZ - grown) * (H + logZ) - grown
    return grown, H


def _column_sums(a, one_by_one):
    """``a.sum(axis=0)``.  numpy adds the rows of a C-contiguous [k, n > 1] array one after the other; with a
    single column the summed axis is the contiguous one and it sums pairwise.  A rank that integrates ONE
    column of an analysis with several (``parallel.LocalColumns``) asks for the first order explicitly."""
    if not one_by_one:
        return a.sum(axis=0)
    total = a[0].copy()
    for row in a[1:]:
        total += row
    return total


def integrate_remainder(sampler, logwidth, logVolremaining, logZ, H, globalLmax):
    """Evidence still held by the live points, which all share the shell width ``logwidth``
    (multi_nested_integrator.py:26-59).  Returns per running data set: remainder log-evidence,
    its bracketing error, total log-evidence, total error (twice, as the reference does)."""
    # ascending likelihood, per data set: only the likelihoods are looked at here, and those are the
    # sorted columns of the live matrix (the reference's `remainder()` also gathers the coordinates
    # of every point: 200 gathers over all data sets per call)
    if hasattr(sampler, 'remainder_likelihoods'):
        live_L = sampler.remainder_likelihoods()
    else:
        live_L = numpy.array([Li for _, _, Li in sampler.remainder()])
    ref = globalLmax                                     # likelihoods are taken relative to this
    rel = numpy.exp(live_L - ref)                        # [nlive, nrunning]
    # upper / lower Riemann sums: every point takes its upper (resp. lower) neighbour's value
    rel_top = rel.copy()
    rel_top[-1] = numpy.exp(globalLmax - ref)
    one_by_one = bool(getattr(sampler, 'sums_row_by_row', False))
    upper = _column_sums(rel_top[1:], one_by_one) + rel_top[-1]
    lower = _column_sums(rel[:-1], one_by_one) + rel[0]
    log_mid = log(_column_sums(rel, one_by_one)) + ref
    total_mid = logaddexp(logZ, logwidth + log_mid)
    total_up = logaddexp(logZ, logwidth + log(upper) + ref)
    total_lo = logaddexp(logZ, logwidth + log(lower) + ref)
    bracket = total_up - total_lo
    assert numpy.isfinite(H).all()
    assert numpy.isfinite(bracket).all(), bracket

    # information if the run stopped here: absorb the live points one by one
    for Li in live_L:
        logZ, H = _absorb(logZ, H, logwidth, Li)
        H[H < 0] = 0

    spread = bracket + (H / sampler.nlive_points) ** 0.5
    return logwidth + log_mid, bracket, total_mid, spread, spread


def multi_nested_integrator(multi_sampler, tolerance=0.01, max_samples=None, min_samples=0,
                            need_robust_remainder_error=True):
    """Run ``multi_sampler`` until, for every data set, the evidence uncertainty (shell
    statistics + live-point remainder) is below ``tolerance`` (checked every 50 iterations,
    multi_nested_integrator.py:136), dropping finished data sets from the sampler as it goes."""
    nlive = multi_sampler.nlive_points
    ntotal = multi_sampler.ndata
    shell = log(1 - exp(-1. / nlive))                    # log width of one shell at volume 1
    log_volume = 0                                       # log prior volume still enclosed
    weights = []

    active = numpy.ones(ntotal, dtype=bool)              # data sets still being sampled
    tail_Z = numpy.zeros(ntotal)                         # remainder evidence at the last check
    tail_Zerr = numpy.zeros(ntotal)
    stat_err = numpy.zeros(ntotal)
    dead_u, dead_x, dead_L = next(multi_sampler)
    logZ = shell + dead_L
    H = dead_L - logZ
    ndim = dead_u.shape[1]
    # live points of each data set at its termination, in order of increasing likelihood
    tail_u = numpy.zeros((nlive, ntotal, ndim))
    tail_x = numpy.zeros((nlive, ntotal, ndim))
    tail_L = numpy.zeros((nlive, ntotal))
    tail_w = numpy.zeros(ntotal)
    it = 0
    while True:
        it += 1
        logwidth = log(1 - exp(-1. / nlive)) + log_volume
        log_volume -= 1. / nlive

        # one weighted sample per data set; finished data sets get zero weight
        row_L = numpy.full(ntotal, -numpy.inf)
        row_L[active] = dead_L
        row_u = numpy.zeros((ntotal, ndim))
        row_u[active, :] = dead_u
        row_x = numpy.zeros((ntotal, ndim))
        row_x[active, :] = dead_x
        weights.append([row_u, row_x, row_L, numpy.where(active, logwidth, -numpy.inf), active])

        stat_err[active] = (H[active] / nlive) ** 0.5

        check = (it > min_samples and it % 50 == 1) or (max_samples and it > max_samples)
        if check:
            remZ, remZerr, _, total_err, _ = integrate_remainder(
                multi_sampler, logwidth, log_volume, logZ[active], H[active], multi_sampler.Lmax)
            tail_Z[active] = remZ
            tail_Zerr[active] = remZerr
            done = total_err < tolerance
            if max_samples and it > max_samples:
                done[:] = True
            # (a sampler whose data sets are spread over ranks takes part in every check: its cut_down is a collective)
            if done.any() or getattr(multi_sampler, 'collective_checks', False):
                log_.debug('iteration %d: %d data sets finished', it, done.sum())
                many = getattr(multi_sampler, 'remainder_arrays_many', None)
                if many is not None:
                    # all finished data sets in one gather (100 000 Python calls at the end of a capped run otherwise)
                    js = numpy.flatnonzero(done)
                    ks = numpy.flatnonzero(active)[js]
                    tail_u[:, ks], tail_x[:, ks], tail_L[:, ks] = many(js)
                    tail_w[ks] = logwidth
                else:
                    for j, k in enumerate(numpy.flatnonzero(active)):
                        if done[j]:
                            tail_u[:, k], tail_x[:, k], tail_L[:, k] = multi_sampler.remainder_arrays(j)
                            tail_w[k] = logwidth
                multi_sampler.cut_down(~done)
                active[active] = ~done
            everybody_done = getattr(multi_sampler, 'everybody_done', None)
            if (not active.any()) if everybody_done is None else everybody_done(not active.any()):
                break
        dead_u, dead_x, dead_L = next(multi_sampler)
        logZ[active], H[active] = _absorb(logZ[active], H[active], logwidth, dead_L)

    # the live points at termination complete the posterior sample (not needed for logZ)
    everyone = numpy.ones(ntotal, dtype=bool)
    for k in range(nlive):
        weights.append([tail_u[k], tail_x[k], tail_L[k], tail_w.copy(), everyone])

    # the reference returns its loop variable after the tail loop, i.e. nlive-1
    # (SURVEY.md appendix A#12); sample.py reports len(weights) instead.  We keep that quirk
    # under the reference's key and add the true count.
    return dict(logZ=logaddexp(logZ, tail_Z), logZerr=stat_err + tail_Zerr, weights=weights,
                information=H, niterations=nlive - 1, nsamples=it)


__all__ = ['multi_nested_integrator', 'integrate_remainder']
