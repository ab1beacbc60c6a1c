"""The MUSE-style problem of BASELINE.json configs[4]: one template fitted to N spectra with
per-pixel variances, amplitude marginalised per spectrum -- the likelihood of the reference's
``musefuse.py:520-535`` (``multi_loglikelihood_clike``: host template, then ``cmuselike.so``),
wired to the same sampler / integrator / constrainers as ``musefuse.py:607-675``.

What differs from the reference's script is the MODEL, by necessity: its stellar-population
template needs external grids and a FITS cube (musefuse.py:31-154,171-284) that are not part of
the repository; SURVEY.md 8(d) defines the stand-in used here and in BASELINE.json configs[4] --
three Gaussian emission lines on a flat continuum, 5 parameters
(:func:`massivedatans_amd.gen.muse_template`).  The LIKELIHOOD is the reference's, including
the ``N(0, 1e-5)`` tie-breaking noise it adds to every evaluation from the global random stream
(musefuse.py:535): with ``jitter=True`` (default) a run consumes the stream exactly as the
reference's loop would (tests/test_muse.py pins that against the reference's own sampler driven
with this problem); ``jitter=False`` drops the noise (kernel benchmarks).

    python -m massivedatans_amd.musefuse <cube.npz> <ndata>
"""
import json
import os
import sys
import time

import numpy

from . import cachedconstrainer, gen
from .multi_nested_integrator import multi_nested_integrator

paramnames = ['log_amp', 'z', 'log_width', 'ratio1', 'ratio3']
nparams = len(paramnames)
#: unit cube -> parameter, per dimension: x = a * u + b
PRIOR = ((2.0, -1.0), (0.02, 0.0), (1.0, -0.5), (1.8, 0.2), (1.8, 0.2))
JITTER_SIGMA = 1e-5                       # musefuse.py:535


def priortransform(cube):
    cube = cube.copy()
    for k, (a, b) in enumerate(PRIOR):
        cube[k] = cube[k] * a + b if b != 0.0 else cube[k] * a
    return cube


def priortransform_batch(cubes):
    cubes = numpy.asarray(cubes, dtype=float)
    out = numpy.empty_like(cubes)
    for k, (a, b) in enumerate(PRIOR):
        out[:, k] = cubes[:, k] * a + b if b != 0.0 else cubes[:, k] * a
    return out


def kernel_params(xs):
    """The device template takes the physical parameters as they are."""
    return numpy.array(xs, dtype=float)


def native_prior(jitter):
    from . import constrainer
    p = constrainer.Prior()
    p.ndim, p.nparams = nparams, nparams
    for k, (a, b) in enumerate(PRIOR):
        p.a[k], p.b[k], p.pow10[k], p.kernel_pow10[k] = a, b, 0, 0
    p.jitter_sigma = JITTER_SIGMA if jitter else 0.0
    return p


class MuseProblem(object):
    """``x`` f64[nx], ``y`` / ``v`` f64[nx, ndata] (the reference's layout, cmuselike.c:54).
    ``backend``: any object with ``loglike_batch(ypred[B, nx], data_mask) -> L[B, M]`` and
    ``loglike_batch_lines(params[B, 5], data_mask)`` (tests inject the CPU oracle there); by
    default :class:`massivedatans_amd.like.MuseSpectra` on the GPU."""

    nparams = nparams
    priortransform = staticmethod(priortransform)
    priortransform_batch = staticmethod(priortransform_batch)

    def __init__(self, x, y, v, backend=None, jitter=True):
        self.x = numpy.ascontiguousarray(x, dtype=float)
        self.y = numpy.ascontiguousarray(y, dtype=float)
        self.v = numpy.ascontiguousarray(v, dtype=float)
        self.nx, self.ndata = self.y.shape
        self.jitter = bool(jitter)
        if backend is None:
            from .like import MuseSpectra
            backend = MuseSpectra(self.x, self.y, self.v)
        self.backend = backend
        self.ncalls = 0
        self.nevals = 0

    def model(self, params):
        return gen.muse_template(self.x, params)

    def multi_loglikelihood(self, params, data_mask):
        """musefuse.py:520-535: template on the host, the C likelihood, the noise."""
        ypred = self.model(params)
        if not numpy.any(ypred):
            return numpy.ones(int(numpy.count_nonzero(data_mask))) * -1e100        # musefuse.py:527-529
        L = self.backend.loglike_batch(ypred[None, :], data_mask)[0]
        self.ncalls += 1
        self.nevals += len(L)
        if self.jitter:
            L = L + numpy.random.normal(0, JITTER_SIGMA, size=len(L))
        return L

    multi_loglikelihood_batch = None          # (every evaluation draws its noise: one candidate at a time)

    def native_prior(self):
        return native_prior(self.jitter)

    def joint_state(self, nlive_points):
        from . import jointstate, parallel
        from .like import MuseSpectra

        def build(scorer, ndata):
            if isinstance(scorer, MuseSpectra):
                return jointstate.MuseJointState(scorer, nlive_points)
            return jointstate.HostJointState(_LinesScorer(scorer), nlive_points, ndata, kernel_params, nparams=nparams)

        if isinstance(self.backend, parallel.ShardedMuse):
            # one process per GPU: every rank keeps the state of ITS block of data sets
            b = self.backend
            js = parallel.ShardedJointState(build(b.local, b.hi - b.lo), self.ndata, b.lo, b.hi)
        else:
            js = build(self.backend, self.ndata)
        js.jitter_sigma = JITTER_SIGMA if self.jitter else 0.0
        return js


class _LinesScorer(object):
    """``loglike_batch(params[B, 5], mask)`` over a backend that scores line parameters."""

    def __init__(self, backend):
        self.backend = backend

    def loglike_batch(self, params, data_mask=None):
        return self.backend.loglike_batch_lines(params, data_mask)


def run(x, y, v, nlive_points=400, nsuperset_draws=10, use_graph=True, max_samples=0, min_samples=0,
        tolerance=0.5, seed=1, backend=None, jitter=True, fused=True, native=None):
    """The whole analysis (musefuse.py:607-648); returns ``(results, sampler, problem, duration)``."""
    from .sample import build_sampler, integrate
    problem = MuseProblem(x, y, v, backend=backend, jitter=jitter)
    start = time.time()
    sampler = build_sampler(problem, nlive_points, nsuperset_draws, use_graph, seed, batched=False, fused=fused, native=native)
    results = integrate(sampler, tolerance, min_samples, max_samples)
    if sampler.native is not None:
        sampler.native.sync_gauss_to_numpy()
    return results, sampler, problem, time.time() - start


def distributed_backend(x, y, v):
    """One process per GPU (torchrun): this rank's block of spectra and variances on its GPU behind
    :class:`parallel.ShardedMuse`; None in a single process (see sample.distributed_backend)."""
    from . import sample
    from .parallel import ShardedMuse
    if sample.distributed_setup() is None:
        return None
    from .like import MuseSpectra
    return ShardedMuse(x, y, v, lambda xs, ys, vs: MuseSpectra(xs, ys, vs))


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 3:
        sys.exit("usage: python -m massivedatans_amd.musefuse <cube.npz with x, y, v> <ndata>")
    ndata = int(argv[2])
    data = gen.load(argv[1], ndata)
    nlive_points = int(os.environ.get('NLIVE_POINTS', '400'))
    results, sampler, problem, duration = run(
        data['x'], data['y'], data['v'], nlive_points=nlive_points, backend=distributed_backend(data['x'], data['y'], data['v']),
        nsuperset_draws=int(os.environ.get('SUPERSET_DRAWS', '10')), use_graph=os.environ.get('USE_GRAPH', '1') == '1',
        max_samples=int(os.environ.get('MAXSAMPLES', 100000)), min_samples=int(os.environ.get('MINSAMPLES', 0)))
    from .sample import write_outputs
    prefix = '%s_full_.out_%d' % (argv[1], ndata)
    if not write_outputs(prefix, results, sampler, duration, ndata):
        return
    print('logZ = %.1f +- %.1f' % (results['logZ'][0], results['logZerr'][0]))
    print('ndraws:', sampler.ndraws, 'niter:', len(results['weights']), 'in %.1f s' % duration)


if __name__ == '__main__':
    main()
