"""The toy problem of the reference's ``sample.py`` -- one Gaussian emission line fitted to N
spectra at once -- wired to the MI355X hot path.

Problem definition surface kept from the reference (sample.py:44-108):
``params``, ``nparams``, ``noise_level``, ``priortransform(cube)``,
``multi_loglikelihood(params, data_mask)``; then the same wiring of constrainers, sampler and
integrator (sample.py:131-197) and the same outputs (sample.py:200-217).  Input and output are
HDF5 through h5py where it is installed (``data_widths_100.hdf5`` works as in the reference);
this image has no h5py, so here both are ``.npz`` with the same dataset names.

    python -m massivedatans_amd.sample data_widths_100.npz 100

Environment knobs as in the reference: NLIVE_POINTS (400), SUPERSET_DRAWS (10), MAXSAMPLES,
MINSAMPLES, USE_GRAPH (1, as in the reference: the grouping of data sets by connected components,
computed on the device; 0: the reference's ``generate_subsets_nograph`` walk, native host code).
Both groupings are pinned bit for bit against runs of the reference's own code -- the graph one
through an igraph stand-in that implements igraph's documented vertex / cluster numbering, which is
therefore the one thing assumed rather than observed (oracle/make_trace.py).  CONSTRAINER must be
MLFRIENDS (the other two draw methods live in third-party ``nestle`` and are outside the accelerated
path).
"""
import json
import logging
import os
import sys
import time

import numpy

from . import _host, cachedconstrainer
from .cachedconstrainer import CachedConstrainer, generate_individual_constrainer
from .multi_nested_integrator import multi_nested_integrator
from .multi_nested_sampler import MultiNestedSampler

log = logging.getLogger("massivedatans_amd")

noise_level = 0.01                    # sample.py:45
params = ['A', 'mu', 'sig']           # sample.py:46
nparams = len(params)


def priortransform(cube):
    """Unit cube -> (A, mu, log10 sig): A log-uniform in [0.01, 1], mu in [400, 800],
    log10 sig in [0, 2] (sample.py:52-58)."""
    cube = cube.copy()
    cube[0] = 10 ** (cube[0] * 2 - 2)
    cube[1] = cube[1] * 400 + 400
    cube[2] = cube[2] * 2
    return cube


def priortransform_batch(cubes):
    """``priortransform`` for the rows of ``cubes[B, 3]``, value for value (the products and sums
    are single IEEE operations either way; the powers go through the same C ``pow`` as the
    scalar ``**``, tests/test_sampler_units.py)."""
    cubes = numpy.asarray(cubes, dtype=float)
    out = numpy.empty_like(cubes)
    out[:, 0] = _host.pow10(cubes[:, 0] * 2 - 2)
    out[:, 1] = cubes[:, 1] * 400 + 400
    out[:, 2] = cubes[:, 2] * 2
    return out


def kernel_params(xs):
    """Rows (A, mu, log10 sig) after the prior transform -> (A, mu, sig) as the kernels take
    them: ``sig = 10**log_sig`` of sample.py:103."""
    p = numpy.array(xs, dtype=float)
    p[:, 2] = _host.pow10(p[:, 2])
    return p


class GaussLineProblem(object):
    """``multi_loglikelihood`` of sample.py:101-108 on the GPU.  ``x`` f64[nx], ``y`` f64[nx, ndata]
    (the reference's layout).  ``backend`` may be any object with
    ``loglike_batch(params[B,3], data_mask) -> L[B, M]`` (tests inject the CPU oracle there);
    by default it is :class:`massivedatans_amd.like.GaussLineSpectra`."""

    def __init__(self, x, y, backend=None):
        self.x = numpy.ascontiguousarray(x, dtype=float)
        self.y = numpy.ascontiguousarray(y, dtype=float)
        self.nx, self.ndata = self.y.shape
        if backend is None:
            from .like import GaussLineSpectra
            backend = GaussLineSpectra(self.x, self.y, noise_level=noise_level)
        self.backend = backend
        self.ncalls = 0
        self.nevals = 0

    def multi_loglikelihood(self, params, data_mask):
        A, mu, log_sig_kms = params
        sig = 10 ** log_sig_kms
        L = self.backend.loglike_batch(numpy.array([[A, mu, sig]]), data_mask)[0]
        self.ncalls += 1
        self.nevals += len(L)
        return L

    def multi_loglikelihood_batch(self, params, data_mask):
        """Rows of ``params`` are (A, mu, log10 sig) after priortransform."""
        # (C pow per row: numpy's vectorised power may differ from the scalar one, which
        # sample.py:103 uses, in the last bit)
        L = self.backend.loglike_batch(kernel_params(params), data_mask)
        self.ncalls += 1
        self.nevals += L.size
        return L


    def joint_state(self, nlive_points):
        """The sampler's floating-point state next to the spectra: on the GPU for the HIP
        backend (accept test and shelf fill there), in numpy over any other
        scorer."""
        from . import jointstate, parallel
        from .like import GaussLineSpectra

        def build(scorer, ndata):
            if isinstance(scorer, GaussLineSpectra):
                return jointstate.GaussJointState(scorer, nlive_points, kernel_params, fetch_rows=False)
            return jointstate.HostJointState(scorer, nlive_points, ndata, kernel_params)

        if isinstance(self.backend, parallel.ShardedGaussLine):
            # one process per GPU: every rank keeps the state of ITS block of data sets
            b = self.backend
            return parallel.ShardedJointState(build(b.local, b.hi - b.lo), self.ndata, b.lo, b.hi)
        return build(self.backend, self.ndata)


def native_context(joint, prior, ndata):
    """The shared part of the native constrainers (massivedatans_amd.constrainer) for a joint
    state, or None where they cannot run (library not built).  With the data sets sharded over
    ranks every rank runs the same constrainer on the same random stream; its chunks go through
    ``ShardedJointState.draw_params``, where the ranks exchange their accept flags."""
    from . import constrainer, parallel
    from .jointstate import GaussJointState, HostJointState
    if not constrainer.available():
        return None
    if isinstance(joint, GaussJointState):
        backend = constrainer.hip_backend(joint)
    elif isinstance(joint, (HostJointState, parallel.ShardedJointState)):
        backend = constrainer.python_backend(joint)
    else:
        return None
    return constrainer.NativeContext(backend, prior, ndata)


def build_sampler(problem, nlive_points=400, nsuperset_draws=10, use_graph=False, seed=1, batched=True,
                  fused=False, native=None, core=None):
    """Constrainers + sampler wired as sample.py:131-194 (CONSTRAINER=MLFRIENDS).  ``fused``: the
    likelihood matrix, the shelves' likelihoods and the thresholds live in a joint state
    (``problem.joint_state``) and whole chunks of candidates are scored and decided there.
    ``native`` (default with ``fused``: on, MDNS_NATIVE_CONSTRAINER=0 turns it off): the
    constrainers are ``constrainer.NativeConstrainer`` objects -- one native call per draw.
    ``core`` (default with ``native``: on, MDNS_NATIVE_CORE=0 turns it off): the sampler is a
    ``core.NativeCoreSampler`` -- the whole integer side of an iteration (passes, grouping,
    constrainer cache, draws, shelves) behind one native call."""
    numpy.random.seed(seed)                                      # sample.py:162
    joint = problem.joint_state(nlive_points) if fused else None
    if native is None:
        native = fused and os.environ.get('MDNS_NATIVE_CONSTRAINER', '1') != '0'
    context = None
    if native:
        from . import constrainer
        prior = problem.native_prior() if hasattr(problem, 'native_prior') else constrainer.sample_py_prior()
        context = native_context(joint, prior, problem.ndata)
    # (the problem definition: sample.py's by default, the problem object's own when it has one)
    prior_fn = getattr(problem, 'priortransform', priortransform)
    prior_batch = getattr(problem, 'priortransform_batch', priortransform_batch)
    ndim = getattr(problem, 'nparams', nparams)
    if context is not None:
        # every constrainer of this sampler: MLFriends with the reference driver's settings
        # (sample.py:133-137), in the library
        cachedconstrainer.generate_fresh_constrainer = lambda: context.fresh_constrainer(
            metriclearner='truncatedscaling', force_shrink=True, rebuild_every=1000, metric_rebuild_every=20)
    else:
        cachedconstrainer.generate_fresh_constrainer = cachedconstrainer.generate_fresh_constrainer_mlfriends
    superset_constrainer = cachedconstrainer.generate_fresh_constrainer()
    cc = CachedConstrainer()
    _, _, individual_draw_constrained = generate_individual_constrainer()
    # the graph variant of the grouping runs on the device when the likelihoods do
    from .jointstate import GaussJointState
    on_device = isinstance(joint, GaussJointState) or isinstance(getattr(joint, 'local', None), GaussJointState)
    device_groups = use_graph and on_device and os.environ.get('MDNS_DEVICE_GROUPS', '1') != '0'
    if core is None:
        core = context is not None and os.environ.get('MDNS_NATIVE_CORE', '1') != '0'
    if core:
        from . import core as core_module
        if context is None or not core_module.available():
            raise RuntimeError("the native sampler core needs the native constrainer (libmdns_host.so)")
        # the constrainers live in the library, with the reference driver's settings (sample.py:133-137)
        return core_module.NativeCoreSampler(
            nlive_points=nlive_points, priortransform=prior_fn,
            multi_loglikelihood=problem.multi_loglikelihood, ndim=ndim, ndata=problem.ndata,
            nsuperset_draws=nsuperset_draws, use_graph=use_graph,
            multi_loglikelihood_batch=getattr(problem, 'multi_loglikelihood_batch', None) if batched else None,
            joint_state=joint, priortransform_batch=prior_batch if fused else None,
            device_groups=device_groups, native=context,
            constrainer_settings=('truncatedscaling', 1000, 20, True))
    sampler = MultiNestedSampler(
        nlive_points=nlive_points, priortransform=prior_fn,
        multi_loglikelihood=problem.multi_loglikelihood, ndim=ndim, ndata=problem.ndata,
        superset_draw_constrained=superset_constrainer.draw_constrained,
        individual_draw_constrained=individual_draw_constrained,
        draw_constrained=cc.get, nsuperset_draws=nsuperset_draws, use_graph=use_graph,
        multi_loglikelihood_batch=getattr(problem, 'multi_loglikelihood_batch', None) if batched else None,
        joint_state=joint, priortransform_batch=prior_batch if fused else None,
        device_groups=device_groups, native=context)
    superset_constrainer.sampler = sampler
    cc.sampler = sampler
    return sampler


def run(x, y, nlive_points=400, nsuperset_draws=10, use_graph=False, max_samples=0, min_samples=0,
        tolerance=0.5, seed=1, backend=None, batched=True, fused=None):
    """The whole analysis; returns ``(results, sampler, problem, duration)``.  ``fused`` defaults to
    True on the GPU (MDNS_FUSED=0 turns it off)."""
    problem = GaussLineProblem(x, y, backend=backend)
    if fused is None:
        from .parallel import ShardedGaussLine
        fused = (backend is None or isinstance(backend, ShardedGaussLine)) and os.environ.get('MDNS_FUSED', '1') != '0'
    start = time.time()
    sampler = build_sampler(problem, nlive_points, nsuperset_draws, use_graph, seed, batched, fused)
    results = integrate(sampler, tolerance, min_samples, max_samples)
    if sampler.native is not None:
        # the cached second deviate of numpy's Gaussian generator travels with the stream the native
        # constrainers stepped: Python code that draws after the run sees the reference's numbers
        sampler.native.sync_gauss_to_numpy()
    return results, sampler, problem, time.time() - start


class _TimedSampler(object):
    """The sampler as the integrator sees it, with the time spent inside ``next`` added up (the rest of a
    run is the integrator's own per-data-set work)."""

    def __init__(self, sampler):
        self.__dict__['_s'] = sampler
        self.__dict__['seconds'] = 0.0

    def __getattr__(self, name):
        return getattr(self._s, name)

    def __setattr__(self, name, value):
        setattr(self._s, name, value)

    def __next__(self):
        t0 = time.perf_counter()
        try:
            return next(self._s)
        finally:
            self.__dict__['seconds'] += time.perf_counter() - t0

    next = __next__


def integrate(sampler, tolerance, min_samples, max_samples):
    """multi_nested_integrator over the sampler -- or, with the data sets spread over ranks (a
    ShardedJointState), every rank over ITS columns (parallel.LocalColumns; MDNS_SHARD_INTEGRATION=0: every
    rank over all of them): ``logZ``, ``logZerr``, ``information`` of all data sets on every rank, ``weights``
    of the rank's own columns ``results['columns']``.  ``results['seconds']``: inside the sampler's ``next`` /
    in the integration around it."""
    joint = getattr(sampler, 'joint', None)
    t0 = time.perf_counter()
    if type(joint).__name__ == 'ShardedJointState' and os.environ.get('MDNS_SHARD_INTEGRATION', '1') != '0':
        from .parallel import LocalColumns
        view = LocalColumns(sampler)
        timed = _TimedSampler(view)
        results = view.gather(multi_nested_integrator(tolerance=tolerance, multi_sampler=timed,
                                                      min_samples=min_samples, max_samples=max_samples))
    else:
        timed = _TimedSampler(sampler)
        results = multi_nested_integrator(tolerance=tolerance, multi_sampler=timed,
                                          min_samples=min_samples, max_samples=max_samples)
    results['seconds'] = dict(sampler=timed.seconds, integration=time.perf_counter() - t0 - timed.seconds)
    return results


def save_results(prefix, results, sampler, duration, ndata):
    """Outputs of sample.py:200-217, same dataset names: ``<prefix>.hdf5`` where h5py is installed
    (what the reference's plotting scripts read), ``<prefix>.npz`` otherwise."""
    from . import gen
    u, x, L, w, mask = list(zip(*results['weights']))
    try:
        import h5py  # noqa: F401
        suffix = '.hdf5'
    except ImportError:
        suffix = '.npz'
    gen.write_datasets(prefix + suffix, dict(
        logZ=results['logZ'], logZerr=results['logZerr'], u=numpy.array(u), x=numpy.array(x),
        L=numpy.array(L), w=numpy.array(w), mask=numpy.array(mask), ndraws=sampler.ndraws))
    with open(prefix + '.stats.json', 'w') as f:
        json.dump(dict(ndraws=int(sampler.ndraws), duration=duration, ndata=int(ndata), niter=len(w),
                       nevals=int(sampler.nevals), seconds=results.get('seconds'),
                       fill_seconds=getattr(sampler, 'fill_seconds', None)), f, indent=4)


def write_outputs(prefix, results, sampler, duration, ndata):
    """save_results for a run that may have been sharded; True on the rank that reports (rank 0)."""
    first = int(os.environ.get('RANK', '0')) == 0
    if 'columns' in results:
        # the evidence integration was sharded: every rank writes the posterior samples of its own data sets
        lo, hi = results['columns']
        part = dict(results, logZ=results['logZ'][lo:hi], logZerr=results['logZerr'][lo:hi])
        save_results('%s.cols%d-%d' % (prefix, lo, hi), part, sampler, duration, hi - lo)
        if first:
            from . import gen
            gen.write_datasets(prefix + '.evidence.npz', dict(logZ=results['logZ'], logZerr=results['logZerr']))
    elif first:
        save_results(prefix, results, sampler, duration, ndata)         # every rank holds the same results
    return first


def distributed_setup():
    """The process group of a torchrun launch (one process per GPU) with the library's kernels and the
    RCCL collectives on ONE stream; returns the backend name, or None in a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and os.environ.get('MDNS_FORCE_DIST') != '1':
        return None
    # torch first: its HIP runtime must be the process's one runtime (see bench.py)
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # MDNS_DIST_BACKEND=gloo exchanges through host memory: lets several ranks share one GPU
    # (RCCL wants one device per rank), e.g. to rehearse the N-rank path on a one-GPU box
    backend = os.environ.get('MDNS_DIST_BACKEND', 'nccl')
    device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    if not dist.is_initialized():
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(backend=backend)
    os.environ.setdefault('MDNS_DEVICE', str(device))
    from . import _lib
    if backend == 'nccl':
        # kernels and RCCL collectives on ONE stream (torch's current one; its default is the null
        # stream, which the library reads as "my own"): the accept flags are reduced on the device
        import ctypes
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        _lib.check(_lib.require_device().mdns_set_stream(ctypes.c_void_p(stream.cuda_stream)), 'mdns_set_stream')
    return backend


def distributed_backend(x, y):
    """One process per GPU (torchrun): every rank runs the same host orchestration from the
    same seed, scores only its contiguous block of spectra on its own GPU and all-gathers the
    likelihood columns over RCCL, so all ranks take identical decisions (the 2-rank CPU test
    reproduces the reference trace bit for bit this way).  Returns None in a single process."""
    if distributed_setup() is None:
        return None
    from .like import GaussLineSpectra
    from .parallel import ShardedGaussLine
    return ShardedGaussLine(x, y, lambda xs, ys: GaussLineSpectra(xs, ys, noise_level=noise_level))


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 3:
        sys.exit("usage: python -m massivedatans_amd.sample <data.npz> <ndata>")
    logging.basicConfig(level=os.environ.get('MDNS_LOG', 'WARNING'))
    from . import gen
    ndata = int(argv[2])
    data = gen.load(argv[1], ndata)
    constrainer_type = os.environ.get('CONSTRAINER', 'MLFRIENDS')
    if constrainer_type != 'MLFRIENDS':
        sys.exit("CONSTRAINER=%s is not available: only MLFRIENDS runs on the accelerated path" % constrainer_type)
    nlive_points = int(os.environ.get('NLIVE_POINTS', '400'))
    use_graph = os.environ.get('USE_GRAPH', '1') == '1'               # the reference's default (sample.py:189)
    backend = distributed_backend(data['x'], data['y'])
    results, sampler, problem, duration = run(
        data['x'], data['y'], nlive_points=nlive_points, backend=backend,
        nsuperset_draws=int(os.environ.get('SUPERSET_DRAWS', '10')),
        use_graph=use_graph,
        max_samples=int(os.environ.get('MAXSAMPLES', 0)), min_samples=int(os.environ.get('MINSAMPLES', 0)))
    if backend is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    prefix = '%s_%s_nlive%d_%d.out8' % (argv[1], constrainer_type, nlive_points, ndata)
    if not write_outputs(prefix, results, sampler, duration, ndata):
        return
    print('logZ = %.1f +- %.1f' % (results['logZ'][0], results['logZerr'][0]))
    print('ndraws:', sampler.ndraws, 'niter:', len(results['weights']), 'likelihood evals:', sampler.nevals,
          'in %.1f s' % duration)


if __name__ == '__main__':
    main()
