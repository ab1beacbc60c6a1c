#!/usr/bin/env python3
"""Constrained-draw likelihood throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload horns|nothing|muse] [--batch B]

Workload horns / nothing (BASELINE.json configs[1] / configs[2]; default horns)
-------------------------------------------------------------------------------
One STEP = one nested-sampling iteration with ONE superset constrained draw, everything
resident in HBM when the timed region starts and every decision taken on the device:

  0. the previous step's advance is taken back (the 10 000 replaced live likelihoods rewritten,
     shelves emptied: extra work, it makes every step the same iteration)
  1. prepare     per data set: lowest live likelihood + slot, shelf purge, threshold
                 (multi_nested_sampler.py:130-143, 438-447)
  2. (N > 1)     RCCL all-gather of the shared live-point pool, K/N points per rank
  3. K6          RadFriends safe radius of the pool, 10 bootstrap rounds (cneighbors.c:125-179);
                 radius and membership threshold finished on the device
  4. K3          membership count of 1000 box candidates (cneighbors.c:95-119)
  5. K1 + accept B = 256 candidate lines x every spectrum of this rank, `any(L > Lmins)` per
                 candidate in the kernel epilogue (clike.c:34-89 + hiermetriclearn.py:193); no
                 likelihood leaves the kernel, one flag per candidate does
  6. (N > 1)     RCCL all-reduce(MAX) of the 256 flags: every rank learns the accepted candidate
  7. commit      the first flagged candidate (the LAST of the batch here: all 256 are scored
                 whichever is accepted): its likelihood row, one fill bit per data set, shelf
                 appends, next thresholds (multi_nested_sampler.py:482-485)
  8. advance     worst live point of every data set replaced by its shelf head (:494-534)
  9. host        fetches the radius (polled mapped memory) and {accepted index, fill bits}
                 (one 1.3 KB copy; the likelihood row stays on the device, as in a real run) --
                 the round trip a real draw makes

`value` = (candidate, spectrum) likelihood evaluations per second over all ranks.  N > 1: weak
scaling, 10 000 spectra per GPU cut from horns(10 000 N), one process per GPU.

The same run also reports (N = 1): `e2e_full` -- the complete analysis of the same spectra to its
termination criterion; `e2e` -- a complete analysis of the same 10 000 spectra capped
at 400 iterations through the real host orchestration, i.e. SURVEY 8(d)'s "evals inside
draw_constrained / wall time of those draws"; `roofline_hbm_regime` -- K1 one pass over 1.6 GB;
`cpu_baseline` -- the reference's own clike.so on the host cores.

Workload muse (BASELINE.json configs[4], one GPU's share: 6 250 spectra x 4096 channels)
-----------------------------------------------------------------------------------------
One STEP = B templates (default 1) scored against the 6 250 spectra with per-pixel variances
(cmuselike.c:45-64): 410 MB of y and 1/v streamed from HBM (beyond the 256 MiB Infinity Cache).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (v_mfma_f64_16x16x4_f64: 1024 FMAs in 64 cycles per SIMD; the probe in tools/probes reaches 69.4)
NDIM = 3
NLIVE = 100
NBOOT = 10                     # clustering/radfriendsregion.py:59
NCAND = 1000                   # clustering/radfriendsregion.py:124
METRIC = "likelihood evals/sec across N datasets per constrained draw"


def priortransform(cube):
    """sample.py:52-58 + sig = 10**log_sig (sample.py:103), vectorised over rows."""
    return np.column_stack([10 ** (cube[:, 0] * 2 - 2), cube[:, 1] * 400 + 400, 10 ** (cube[:, 2] * 2)])


#: the GPU pool gives a one-GPU job a 16-core share of its host, whatever the host has and the
#: affinity mask says; OpenMP legs use at most that many threads (256 threads on a shared host
#: measured 30x SLOWER than 16)
CORE_SHARE = 16


def host_cores():
    """Threads the OpenMP legs use (the job's core share), and what lscpu says about the host."""
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    usable = min(usable, CORE_SHARE)
    info = {}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in ("Model name", "CPU(s)", "Thread(s) per core", "Core(s) per socket", "Socket(s)"):
                info[k.strip()] = v.strip()
    except Exception:      # noqa: BLE001
        pass
    return usable, info


def ref_provenance():
    from oracle.oracle import have_reference
    if have_reference():
        return ("oracle/_ref/*.so: the reference's own C compiled by oracle/Makefile where /root/reference exists "
                "(the build container), shipped prebuilt to this box -- NOT built here")
    return "oracle/liboracle.so: our C restatement (bit-identical to the reference C in tests/test_oracle.py)"


def cpu_baseline_gauss(data, params, budget_s=12.0):
    """Reference CPU path for K1 on the host cores of this box, bounded sample.  Serial, like
    the reference (sample.py:81 never loads clike-parallel)."""
    from oracle.oracle import Oracle, have_reference
    kind = "reference" if have_reference() else "port"
    orc = Oracle(kind)
    x, y = data["x"], data["y"]
    nd = y.shape[1]
    mask = np.ones(nd, dtype=np.bool_)
    out = np.zeros(nd)
    orc.gauss_like(x, y, params[0, 0], params[0, 1], params[0, 2], 0.01, mask, Lout=out)   # warm
    t0 = time.perf_counter()
    n = 0
    while True:
        p = params[n % len(params)]
        out[:] = 0
        orc.gauss_like(x, y, p[0], p[1], p[2], 0.01, mask, Lout=out)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 20000:
            break
    usable, info = host_cores()
    res = {"value": n * nd / el, "unit": "likelihood evals/s", "cores": 1, "kind": kind,
           "sample": "%d candidates x %d spectra x 200 channels, full mask, serial clike loop, %.1f s" % (n, nd, el),
           "binary": ref_provenance(), "host": info, "core_share": usable}
    # best-effort multi-threaded form (dataset-parallel OpenMP restatement), for context only
    try:
        os.environ.setdefault("OMP_NUM_THREADS", str(usable))
        omp = Oracle("port-omp")
        t0 = time.perf_counter()
        m = 0
        while time.perf_counter() - t0 < 4.0:
            p = params[m % len(params)]
            out[:] = 0
            omp.gauss_like(x, y, p[0], p[1], p[2], 0.01, mask, Lout=out)
            m += 1
        res["value_openmp"] = m * nd / (time.perf_counter() - t0)
        res["cores_openmp"] = int(os.environ.get("OMP_NUM_THREADS", usable))
    except Exception as e:      # noqa: BLE001
        res["openmp_error"] = str(e)
    return res


def hbm_regime_leg(lib, _lib, ndata=1000000, nx=200, reps=20):
    """The one-pass (B = 1) row kernel on spectra that do NOT fit the 256 MiB Infinity Cache:
    1 000 000 x 200 doubles = 1.6 GB streamed from HBM per launch.  This is the HBM-bound
    regime of K1 (1608 algorithmic bytes per eval = the bytes physically moved)."""
    rng = np.random.RandomState(ndata)
    x = np.linspace(400, 800, nx)
    y = np.ascontiguousarray(rng.normal(0, 0.01, size=(ndata, nx)))
    sp = lib.mdns_spectra_create(_lib.ptr(x), _lib.ptr(y), None, ndata, nx, 1)
    if not sp:
        return {"error": _lib.last_error()}
    del y
    p = np.array([[0.3, 640.0, 5.0]])
    d_p = lib.mdns_dev_alloc(p.nbytes)
    lib.mdns_h2d(d_p, _lib.ptr(p), p.nbytes)
    d_L = lib.mdns_dev_alloc(ndata * 8)
    for _ in range(3):
        lib.mdns_gauss_loglike_batch_dev(sp, d_p, 1, 0.01, None, ndata, d_L)
    lib.mdns_sync()
    lib.mdns_profile(1)
    for _ in range(reps):
        lib.mdns_gauss_loglike_batch_dev(sp, d_p, 1, 0.01, None, ndata, d_L)
    lib.mdns_sync()
    n, ms = C.c_longlong(0), C.c_double(0)
    lib.mdns_profile_read(0, C.byref(n), C.byref(ms))
    lib.mdns_profile(0)
    us = 1e3 * ms.value / max(1, n.value)
    lib.mdns_dev_free(d_p)
    lib.mdns_dev_free(d_L)
    lib.mdns_spectra_destroy(sp)
    gbs = ndata * (8 * nx + 8) / (us * 1e-6) / 1e9
    return {"kernel": (lib.mdns_profile_kernel(0) or b"").decode(), "workload": "%d spectra x %d channels (%.1f GB > 256 MiB Infinity Cache), "
                                                  "1 candidate per pass" % (ndata, nx, ndata * nx * 8 / 1e9),
            "bound": "hbm", "launch_us": us, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "evals_per_s": ndata / (us * 1e-6)}


def muse_leg(lib, _lib, nd=6250, nx=4096):
    """K2 (cmuselike.c:34-66) on one GPU's share of BASELINE configs[4], inside the default line: one
    template per pass (the HBM regime: y and 1/v, 410 MB, streamed once per launch) and 64 templates per
    pass (fp64 vector issue); HIP events around every launch, HBM traffic from the committed --pmc passes."""
    from massivedatans_amd import gen
    cube = gen.muse_like(nd, nx)
    x = cube["x"]
    sp = lib.mdns_spectra_create(_lib.ptr(x), _lib.ptr(np.ascontiguousarray(cube["y"])), _lib.ptr(np.ascontiguousarray(cube["v"])), nd, nx, 0)
    if not sp:
        return {"error": _lib.last_error()}
    rng = np.random.RandomState(3)
    out = {"workload": "%d spectra x %d channels with per-pixel variances (y and 1/v = %.0f MB > 256 MiB Infinity Cache)"
                       % (nd, nx, 16 * nx * nd / 1e6)}
    for B, reps in ((1, 30), (64, 8)):
        p5 = np.column_stack([rng.uniform(-0.3, 0.3, B), rng.uniform(0.0, 0.02, B), rng.uniform(-0.2, 0.2, B),
                              rng.uniform(0.5, 1.5, B), rng.uniform(0.5, 1.5, B)])
        templates = np.array([gen.muse_template(x, p) for p in p5])
        d_t = lib.mdns_dev_alloc(templates.nbytes)
        lib.mdns_h2d(d_t, _lib.ptr(templates), templates.nbytes)
        d_L = lib.mdns_dev_alloc(B * nd * 8)
        for _ in range(2):
            lib.mdns_muse_loglike_batch_dev(sp, d_t, B, None, nd, d_L)
        lib.mdns_sync()
        lib.mdns_profile_every(1)
        lib.mdns_profile(2)
        for _ in range(reps):
            lib.mdns_muse_loglike_batch_dev(sp, d_t, B, None, nd, d_L)
        lib.mdns_sync()
        n, us = read_profile(lib, 1)
        lib.mdns_profile(0)
        kernel = (lib.mdns_profile_kernel(1) or b"").decode()
        lib.mdns_dev_free(d_t)
        lib.mdns_dev_free(d_L)
        phys = 16 * nx * nd + 8 * B * nd + 8 * B * nx
        traffic, source = pmc_traffic(kernel)
        if B == 1:
            gbs = phys / (us * 1e-6) / 1e9
            out["b1"] = {"bound": "hbm", "kernel": kernel, "launch_us": us, "launches_timed": n, "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": phys, "traffic": traffic,
                         "traffic_source": source, "evals_per_s": nd / (us * 1e-6)}
        else:
            tf = 10.0 * nx * B * nd / (us * 1e-6) / 1e12
            out["b64"] = {"bound": "fp64_valu", "kernel": kernel, "launch_us": us, "launches_timed": n, "achieved": tf,
                          "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
                          "flops_per_launch": 10.0 * nx * B * nd, "traffic": traffic, "traffic_source": source,
                          "evals_per_s": B * nd / (us * 1e-6)}
    # the accept pass of a 64-candidate chunk as two matrix products (csrc/mdns_k2gemm.hip): what a constrained
    # draw runs for chunks of this size; thresholds far from the candidates, a noise bound as in a run
    B = 64
    p5 = np.column_stack([rng.uniform(-0.3, 0.3, B), rng.uniform(0.0, 0.02, B), rng.uniform(-0.2, 0.2, B),
                          rng.uniform(0.5, 1.5, B), rng.uniform(0.5, 1.5, B)])
    templates = np.array([gen.muse_template(x, p) for p in p5])
    d_t = lib.mdns_dev_alloc(templates.nbytes)
    lib.mdns_h2d(d_t, _lib.ptr(templates), templates.nbytes)
    thr, bound, flags = np.full(nd, 1e30), np.full(B, 5e-5), np.zeros(2 * B + 1, dtype=np.int32)
    d_thr, d_bound, d_flags = lib.mdns_dev_alloc(thr.nbytes), lib.mdns_dev_alloc(bound.nbytes), lib.mdns_dev_alloc(flags.nbytes)
    lib.mdns_h2d(d_thr, _lib.ptr(thr), thr.nbytes)
    lib.mdns_h2d(d_bound, _lib.ptr(bound), bound.nbytes)
    lib.mdns_h2d(d_flags, _lib.ptr(flags), flags.nbytes)
    for _ in range(2):
        lib.mdns_muse_filter_dev(sp, d_t, B, None, nd, d_thr, d_bound, d_flags)
    lib.mdns_sync()
    lib.mdns_profile_every(1)
    lib.mdns_profile(2)
    for _ in range(20):
        lib.mdns_muse_filter_dev(sp, d_t, B, None, nd, d_thr, d_bound, d_flags)
    lib.mdns_sync()
    n, us = read_profile(lib, 1)
    lib.mdns_profile(0)
    kernel = (lib.mdns_profile_kernel(1) or b"").decode()
    lib.mdns_d2h(_lib.ptr(flags), d_flags, flags.nbytes)
    for d in (d_t, d_thr, d_bound, d_flags):
        lib.mdns_dev_free(d)
    nxp = (nx + 15) // 16 * 16
    tf = 4.0 * nxp * B * nd / (us * 1e-6) / 1e12
    traffic, source = pmc_traffic(kernel)
    out["b64_filter"] = {"bound": "mfma", "kernel": kernel, "launch_us": us, "launches_timed": n, "achieved": tf,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                         "flops_per_launch": 4.0 * nxp * B * nd, "bytes_per_launch": 16 * nxp * nd + 8 * B * nxp, "traffic": traffic,
                         "traffic_source": source, "evals_per_s": B * nd / (us * 1e-6), "undecided_pairs": int(flags[2 * B]),
                         "note": "executed flops: 2 products x 2 per (candidate, channel, spectrum) on v_mfma_f64_16x16x4_f64; "
                                 "decisions are the exact kernel's (tests/test_muse.py), the accepted candidate's row is made by it"}
    lib.mdns_spectra_destroy(sp)
    return out


def cpu_baseline_e2e(data, iterations):
    """The same analysis -- our host orchestration, native constrainer, USE_GRAPH=1 -- on the CPU oracle's
    kernels (the reference's own C where oracle/_ref is built), capped after `iterations` iterations
    (about 15 s of one host core): the like-for-like CPU number for the constrained-draw metric.  Runs in
    the cpu_baseline leg only, after the timed region; the geometry entry points of the package are
    pointed at the oracle for its duration and put back."""
    from massivedatans_amd import sample
    from massivedatans_amd.clustering import neighbors
    from oracle.oracle import Oracle, have_reference
    from oracle.backends import OracleSpectra, patch_neighbors
    kind = "reference" if have_reference() else "port"
    orc = Oracle(kind)
    saved = {}

    class _Patch(object):
        def setattr(self, obj, name, value):
            saved.setdefault((obj, name), getattr(obj, name))
            setattr(obj, name, value)

    patch_neighbors(_Patch(), orc)
    try:
        with np.errstate(all="ignore"):
            results, sampler, problem, duration = sample.run(data["x"], data["y"], nlive_points=NLIVE, max_samples=iterations,
                                                             use_graph=True, backend=OracleSpectra(orc, data["x"], data["y"]),
                                                             fused=True)
    finally:
        for (obj, name), value in saved.items():
            setattr(obj, name, value)
    evals_draws = int(sampler.nevals) - NLIVE * data["y"].shape[1]
    return {"workload": "complete analysis of the same spectra on the CPU oracle's kernels, capped at %d iterations" % iterations,
            "kind": kind, "cores": 1, "wall_s": duration, "iterations": int(results["nsamples"]), "ndraws": int(sampler.ndraws),
            "constrained_draws": int(sampler.ndraw_calls), "evals_useful": evals_draws,
            "draw_constrained_wall_s": sampler.draw_seconds,
            "evals_per_s_in_draw_constrained": evals_draws / sampler.draw_seconds if sampler.draw_seconds else None,
            "evals_per_s_whole_run": int(sampler.nevals) / duration, "logZ_first3": [float(v) for v in results["logZ"][:3]]}


def e2e_leg(data, iterations, use_graph=False):
    """SURVEY 8(d)'s metric as it is defined: a real analysis (sampler + integrator + constrainers
    on the host, every kernel and the accept / fill decisions on the device) of the same spectra,
    capped after `iterations` nested-sampling iterations; useful (candidate, data set) evaluations
    of the constrained draws divided by the wall-clock spent inside draw_constrained.
    use_graph: the reference's default grouping (connected components), computed on the device;
    otherwise the pinned walk on the host (sample.py:189 USE_GRAPH)."""
    from massivedatans_amd import sample
    with np.errstate(all="ignore"):
        results, sampler, problem, duration = sample.run(data["x"], data["y"], nlive_points=NLIVE,
                                                         max_samples=iterations, use_graph=use_graph)
    joint = sampler.joint
    evals_draws = int(sampler.nevals) - NLIVE * data["y"].shape[1]          # without the initial live points
    out = {"workload": "complete analysis of the same spectra, capped at %d iterations (tolerance 0.5, nlive %d)"
                       % (iterations, NLIVE),
           "wall_s": duration, "iterations": int(results["nsamples"]), "ndraws": int(sampler.ndraws),
           "constrained_draws": int(sampler.ndraw_calls), "draw_chunks": int(sampler.ndraw_chunks),
           "launch_sequences": int((joint.ncalls if joint is not None else 0) + problem.ncalls),
           "evals_useful": evals_draws,
           "evals_scored": int((joint.nevals_scored if joint is not None else 0) + problem.nevals),
           "draw_constrained_wall_s": sampler.draw_seconds,
           "evals_per_s_in_draw_constrained": evals_draws / sampler.draw_seconds if sampler.draw_seconds else None,
           "evals_per_s_whole_run": int(sampler.nevals) / duration,
           "fused": joint is not None, "logZ_first3": [float(v) for v in results["logZ"][:3]],
           "grouping": ("connected components on the device (USE_GRAPH=1), %d calls, %.1f rounds each"
                        % (sampler._dgroups.ncalls, sampler._dgroups.mean_rounds())) if sampler._dgroups is not None
           else ("connected components on the host (USE_GRAPH=1)" if use_graph else "discovery-order walk on the host (USE_GRAPH=0)")}
    if sampler.native is not None:
        # the constrained draws ran in the library (one native call per draw): its own counters
        st = sampler.native.stats()
        out["native_constrainer"] = st
        out["draw_chunks"] = int(st["chunks"])
        out["launch_sequences"] = int(st["chunks"] + st["radii"] + st["counts"])
        out["evals_scored"] = int(st["pairs"] + NLIVE * data["y"].shape[1])
    if hasattr(sampler, "core_stats"):
        # the integer side of every iteration ran in the library as well (mdns.h Part 6): its counters
        out["core"] = sampler.core_stats()
    if iterations == 0:
        out["workload"] = "COMPLETE analysis of the same spectra to the termination criterion (tolerance 0.5, nlive %d)" % NLIVE
    if joint is not None:
        joint.close()
    return out


def e2e_muse_leg(x, y_t, v_t, iterations, jitter):
    """The same for the MUSE-style problem on this GPU's spectra: musefuse.py:520-535 behind the
    sampler (massivedatans_amd.musefuse), K2 joint state on the device, native constrainer.
    jitter: the N(0, 1e-5) noise of musefuse.py:535 drawn on the host from the global stream, per
    evaluated candidate and data set, as the reference does (False: without)."""
    from massivedatans_amd import musefuse
    with np.errstate(all="ignore"):
        results, sampler, problem, duration = musefuse.run(x, y_t, v_t, nlive_points=NLIVE, max_samples=iterations, jitter=jitter)
    nd = y_t.shape[1]
    evals_draws = int(sampler.nevals) - NLIVE * nd
    st = sampler.native.stats() if sampler.native is not None else {}
    out = {"workload": "complete MUSE-style analysis of these %d spectra x %d channels, capped at %d iterations (nlive %d), "
                       "likelihood noise %s" % (nd, y_t.shape[0], iterations, NLIVE, "on (musefuse.py:535)" if jitter else "off"),
           "wall_s": duration, "iterations": int(results["nsamples"]), "ndraws": int(sampler.ndraws),
           "constrained_draws": int(sampler.ndraw_calls), "evals_useful": evals_draws,
           "draw_constrained_wall_s": sampler.draw_seconds,
           "evals_per_s_in_draw_constrained": evals_draws / sampler.draw_seconds if sampler.draw_seconds else None,
           "evals_per_s_whole_run": int(sampler.nevals) / duration, "native_constrainer": st,
           "logZ_first3": [float(v) for v in results["logZ"][:3]]}
    if sampler.joint is not None:
        sampler.joint.close()
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this command
    (profiles/r*_pmc.json, newest round first: FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md
    HBM section; tools/collect_profiles.sh + tools/profile_summary.py make them)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            doc = json.load(open(path))
            entry = doc["kernels"].get("mdns::" + kernel)
            if entry is not None:
                return entry["hbm_bytes"], "profiles/%s (committed rocprofv3 --pmc passes of this command)" % os.path.basename(path)
        except Exception:      # noqa: BLE001
            continue
    return None, None


# ------------------------------------------------------------------------------------------
COMM = {"direct": None, "rccl": None, "stream": None}      # set by setup_dist in a distributed run


def setup_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    torch = dist = None
    # MDNS_BENCH_FORCE_DIST=1 exercises the collective path with a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("MDNS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        # torch first: it brings its own copy of the HIP runtime (same soname); loaded first it
        # becomes the one runtime of the process, which libmdns_hip.so then binds to as well
        import torch
        import torch.distributed as dist
        device_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(device_index)
        if world == 1:
            # (the single-rank rehearsal started by hand: what torchrun would have put there)
            for key, value in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29611")):
                os.environ.setdefault(key, value)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
    else:
        device_index = local_rank
    from massivedatans_amd import _lib
    os.environ.setdefault("MDNS_DEVICE", str(device_index))
    lib = _lib.require_device()
    if use_dist:
        # The kernels go on torch's CURRENT stream, which the (synchronous-API) RCCL collectives
        # order themselves against on both sides.  A stream of our own, because torch's default
        # stream is the null stream (handle 0), which mdns_set_stream reads as "library stream".
        bench_stream = torch.cuda.Stream()
        torch.cuda.set_stream(bench_stream)
        assert bench_stream.cuda_stream != 0
        _lib.check(lib.mdns_set_stream(C.c_void_p(bench_stream.cuda_stream)), "mdns_set_stream")
        # The exchanges of a step are tiny (B flags, one bit per data set, the pool): RCCL is
        # called directly on that stream (massivedatans_amd/rccl.py), one enqueue per collective;
        # MDNS_BENCH_COLLECTIVES=torch goes through torch.distributed instead.
        COMM["stream"] = bench_stream.cuda_stream
        if os.environ.get("MDNS_BENCH_COLLECTIVES", "rccl") != "torch":
            from massivedatans_amd import rccl
            try:
                comm = rccl.from_torch_distributed()
            except (rccl.RcclError, OSError, AttributeError) as e:
                comm = None
                print("bench.py: direct RCCL unavailable (%s), using torch.distributed" % e, file=sys.stderr)
            # one tiny MAX all-reduce and one all-gather, checked on every rank; the direct path is
            # used only if they came out right everywhere (decided through torch.distributed)
            ok = 0
            if comm is not None:
                try:
                    a = torch.full((8,), rank + 1, dtype=torch.int32, device="cuda")
                    b = torch.full((3,), 7 * rank + 5, dtype=torch.int64, device="cuda")
                    g = torch.zeros(3 * world, dtype=torch.int64, device="cuda")
                    comm.all_reduce(a.data_ptr(), a.data_ptr(), 8, rccl.INT32, rccl.MAX, bench_stream.cuda_stream)
                    comm.all_gather(b.data_ptr(), g.data_ptr(), 3, rccl.INT64, bench_stream.cuda_stream)
                    bench_stream.synchronize()
                    want = torch.arange(world, dtype=torch.int64).repeat_interleave(3) * 7 + 5
                    ok = int(bool((a == world).all().item()) and bool((g.cpu() == want).all().item()))
                except rccl.RcclError as e:
                    print("bench.py: direct RCCL self-test failed (%s)" % e, file=sys.stderr)
            agree = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 1:
                COMM["rccl"], COMM["direct"] = rccl, comm
            elif rank == 0:
                print("bench.py: direct RCCL self-test not passed on every rank, using torch.distributed", file=sys.stderr)
    return world, rank, use_dist, torch, dist, lib, _lib


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks of this script through
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) BEFORE this process has touched
    the GPU, relay what they print -- rank 0 prints the JSON line -- and return their exit code.
    (Never an exec: a process that has initialised HIP must not be replaced, and this one has not
    and does not.)"""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def timed(step, fence, args, use_dist, torch, dist, lib, _lib):
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(lib.mdns_sync(), "sync")
    if use_dist:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def read_profile(lib, which):
    n, ms = C.c_longlong(0), C.c_double(0)
    lib.mdns_profile_read(which, C.byref(n), C.byref(ms))
    return int(n.value), (1e3 * ms.value / n.value if n.value else 0.0)


def bench_gauss(args):
    world, rank, use_dist, torch, dist, lib, _lib = setup_dist(args)
    from massivedatans_amd import gen

    # ---- synthetic input, resident before the timed region --------------------------------
    nd, B, K = args.ndata, args.batch, args.pool
    if B < 2 or B > _lib.JOINT_MAX_BATCH:
        sys.exit("--batch must be in [2, %d]" % _lib.JOINT_MAX_BATCH)
    make = gen.horns if args.workload == "horns" else gen.nothing
    full = make(nd * world)
    shard = np.ascontiguousarray(full["y"][:, rank * nd:(rank + 1) * nd])
    data = {"x": full["x"], "y": shard}
    nx = shard.shape[0]
    spectra = lib.mdns_spectra_create(_lib.ptr(data["x"]), _lib.ptr(shard), None, nd, nx, 0)
    if not spectra:
        raise _lib.MdnsError(_lib.last_error())
    joint = lib.mdns_joint_create(spectra, NLIVE, 8)
    if not joint:
        raise _lib.MdnsError(_lib.last_error())

    rng = np.random.RandomState(1)                      # sample.py:162
    live_params = priortransform(rng.uniform(size=(NLIVE, NDIM)))
    _lib.check(lib.mdns_joint_init_gauss(joint, _lib.ptr(live_params), 0.01), "init")
    d_saved = lib.mdns_dev_alloc(NLIVE * nd * 8)           # the initial matrix, for the check after the timed region
    _lib.check(lib.mdns_d2d(d_saved, lib.mdns_joint_live_dev(joint), NLIVE * nd * 8), "save live")
    # B - 1 candidates no data set accepts (lines ten times brighter than the prior allows, as
    # broad as it allows), then one that beats the worst live point of every data set (a line at
    # the faint end of the prior) -- checked after the timed region
    params = np.column_stack([np.full(B, 10.0), rng.uniform(400, 800, size=B), np.full(B, 100.0)])
    params[B - 1] = (0.01, 790.0, 1.0)
    pool = rng.uniform(0.3, 0.7, size=(K, NDIM))        # live points (unit cube coordinates)
    chosen = np.zeros((K, NBOOT))
    for b in range(NBOOT):
        chosen[rng.choice(np.arange(K), size=K, replace=True), b] = 1.
    cands = rng.uniform(0.25, 0.75, size=(NCAND, NDIM))

    def dev(a):
        a = np.ascontiguousarray(a)
        p = lib.mdns_dev_alloc(a.nbytes)
        if not p:
            raise _lib.MdnsError(_lib.last_error())
        _lib.check(lib.mdns_h2d(p, _lib.ptr(a), a.nbytes), "h2d")
        return p

    d_params, d_chosen, d_cands = dev(params), dev(chosen), dev(cands)
    d_counts = lib.mdns_dev_alloc(NCAND * 4)
    nres = lib.mdns_joint_result_bytes(nd)
    result = np.zeros(nres, dtype=np.uint8)
    state = {"advanced": False}
    accepted_now = C.c_int(-2)
    fill_now = np.zeros((nd + 63) // 64, dtype=np.uint64)
    d_flags = lib.mdns_joint_flags_dev(joint)
    d_result = lib.mdns_joint_result_dev(joint)
    if use_dist:
        share = (K + world - 1) // world                 # the last rank may hold fewer: padded gather
        t_pool = torch.zeros((world * share, NDIM), dtype=torch.float64, device="cuda")
        mine = np.zeros((share, NDIM))
        part = pool[rank * share:(rank + 1) * share]
        mine[:len(part)] = part
        mine[len(part):] = pool[0]                       # padding repeats a pool point: same region
        t_mine = torch.from_numpy(mine).cuda()
        d_pool = t_pool.data_ptr()
        Kdev = world * share
        # the bootstrap choice must cover the padded pool
        if Kdev != K:
            ch = np.zeros((Kdev, NBOOT))
            ch[:K] = chosen
            d_chosen = dev(ch)
        # the joint state's own buffers, reduced / gathered in place (no staging copies)
        from massivedatans_amd.parallel import device_view
        nbits = (nd + 63) // 64
        t_flags = device_view(d_flags, (B,), "<i4")
        t_bits = device_view(d_result + 16, (nbits,), "<i8")
        t_allbits = torch.zeros(world * nbits, dtype=torch.int64, device="cuda")
        allbits = np.zeros(world * nbits, dtype=np.int64)
    else:
        d_pool = dev(pool)
        Kdev = K
    direct, rccl, stream = COMM["direct"], COMM["rccl"], COMM["stream"]
    region = lib.mdns_region_wrap_dev(d_pool, Kdev, NDIM)     # the pool buffer is refilled in place
    if not region:
        raise _lib.MdnsError(_lib.last_error())

    def step(with_row=False):
        if state["advanced"]:
            _lib.check(lib.mdns_joint_undo_advance_dev(joint), "undo advance")
        if direct is not None:
            direct.all_gather(t_mine.data_ptr(), t_pool.data_ptr(), share * NDIM, rccl.FLOAT64, stream)
        elif use_dist:
            # the pool exchange runs on RCCL's stream beside the state kernels; K6 waits for it
            gathered = dist.all_gather_into_tensor(t_pool, t_mine, async_op=True)
        _lib.check(lib.mdns_joint_prepare_dev(joint), "prepare")
        if use_dist and direct is None:
            gathered.wait()
        # K6, then radius + membership threshold finished on the device: K3 follows in stream order
        _lib.check(lib.mdns_region_bootstrap_radius_async(region, d_chosen, NBOOT), "K6")
        _lib.check(lib.mdns_region_count_dev(region, d_cands, NCAND, d_counts), "K3")
        _lib.check(lib.mdns_joint_score_dev(joint, d_params, B, 0.01, None, nd), "K1 + accept")
        # every rank learns which candidates ANY rank's data sets accept: B flags, not L[B, M]
        if direct is not None:
            direct.all_reduce(d_flags, d_flags, B, rccl.INT32, rccl.MAX, stream)
        elif use_dist:
            dist.all_reduce(t_flags, op=dist.ReduceOp.MAX)
        # (the sampler needs the index and the fill bits of a draw, not its likelihood row: the
        # commit takes both from what the accept pass kept of the candidates it flagged)
        _lib.check((lib.mdns_joint_commit_dev if with_row else lib.mdns_joint_commit_bits_dev)(joint, None, nd), "commit")
        _lib.check(lib.mdns_joint_advance_dev(joint), "advance")
        state["advanced"] = True
        # the host bookkeeping of every rank needs the fill bits of all data sets
        if direct is not None:
            direct.all_gather(d_result + 16, t_allbits.data_ptr(), nbits, rccl.INT64, stream)
        elif use_dist:
            gathered = dist.all_gather_into_tensor(t_allbits, t_bits, async_op=True)
        # the host needs the radius (bounding box of the next proposals) and the outcome of the draw
        radius = lib.mdns_region_radius(region)
        if radius != radius:
            raise _lib.MdnsError(_lib.last_error())
        # ({accepted, status, fill words} arrive in mapped host memory: mdns_joint_fetch polls for them)
        if use_dist:
            if direct is None:
                gathered.wait()
            _lib.check(lib.mdns_joint_fetch(joint, nd, C.byref(accepted_now), None), "outcome")
            _lib.check(lib.mdns_d2h(_lib.ptr(allbits), C.c_void_p(t_allbits.data_ptr()), allbits.nbytes), "fill bits of all ranks")
        else:
            _lib.check(lib.mdns_joint_fetch(joint, nd, C.byref(accepted_now), _lib.ptr(fill_now)), "outcome")

    def fence():
        _lib.check(lib.mdns_sync(), "sync")
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # events only around the dominant kernel (every timed launch adds two event records), and its
    # launches are sampled (every 4th), which keeps the cost of measuring out of `value`
    lib.mdns_profile_every(args.event_every)
    lib.mdns_profile(1 if not args.no_events else 0)
    elapsed = timed(step, fence, args, use_dist, torch, dist, lib, _lib)
    n_launch, k1_us = read_profile(lib, 0)
    kernel = (lib.mdns_profile_kernel(0) or b"").decode()      # e.g. "k_gauss_cols_accept<8>"
    # the geometry kernels of the step, timed in a few extra (unreported) steps
    lib.mdns_profile_every(1)
    lib.mdns_profile(4 | 8)
    for _ in range(min(args.steps, 50)):
        step()
    other = {"count_within_us": read_profile(lib, 2)[1], "bootstrap_us": read_profile(lib, 3)[1]}
    lib.mdns_profile(0)
    fence()

    # sanity: the timed steps decided what they were built to decide, with the right numbers
    # (one more step that also writes the likelihood row, which is recomputed for the check)
    polled_accepted, polled_fill = accepted_now.value, fill_now.copy()
    step(with_row=True)
    fence()
    assert accepted_now.value == polled_accepted and np.array_equal(fill_now, polled_fill), \
        "bench draw: the commit with and without the likelihood row disagree"
    _lib.check(lib.mdns_d2h(_lib.ptr(result), d_result, nres), "result")       # now with the likelihood row
    live_now = np.empty((NLIVE, nd))
    _lib.check(lib.mdns_joint_undo_advance_dev(joint), "undo advance")
    _lib.check(lib.mdns_joint_get_live(joint, _lib.ptr(live_now)), "get live")
    live_first = np.empty((NLIVE, nd))
    _lib.check(lib.mdns_d2h(_lib.ptr(live_first), d_saved, live_first.nbytes), "d2h")
    assert np.array_equal(live_now, live_first), "taking back the advance did not restore the live matrix"
    accepted = int(result[:4].view(np.int32)[0])
    status = int(result[4:8].view(np.int32)[0])
    nbw = (nd + 63) // 64
    bits = np.unpackbits(result[16:16 + 8 * nbw], bitorder="little")[:nd]
    Lrow = result[16 + 8 * nbw:].view(np.float64)
    ypred = params[B - 1, 0] * np.exp(-0.5 * ((params[B - 1, 1] - data["x"]) / params[B - 1, 2]) ** 2)
    want = -0.5 * (((ypred.reshape((-1, 1)) - shard[:, :64]) / 0.01) ** 2).sum(axis=0)
    assert accepted == B - 1 and status == 0, ("bench draw: accepted %d status %d" % (accepted, status))
    assert accepted_now.value == B - 1, "bench draw: the polled outcome differs from the result buffer"
    if not use_dist:
        assert np.array_equal(fill_now.view(np.uint8), result[16:16 + 8 * nbw]), "bench draw: polled fill bits differ"
    assert bits.all(), "bench draw: the accepted candidate must fill every shelf"
    if use_dist:
        every = np.unpackbits(allbits.view(np.uint8).reshape(world, -1), axis=1, bitorder="little")[:, :nd]
        assert every.all(), "bench draw: the gathered fill bits of some rank are incomplete"
    assert np.allclose(Lrow[:64], want, rtol=1e-10), "bench output check failed"

    if rank == 0:
        evals_per_step = B * nd * world
        value = evals_per_step * args.steps / elapsed
        mfma = kernel.startswith("k_gauss_mfma_filter") or kernel.startswith("k_gauss_gemm_filter")
        # per launch.  Chain kernels: subtract, multiply, add per (channel, candidate, spectrum).  The
        # matrix-core filter EXECUTES one multiply-add per element (the cross term of the expanded
        # square): `achieved` counts those 2 flops; the 3 flops of the reference's formula that they
        # stand for are reported beside it, never as the roofline fraction.
        # (k_gauss_gemm_filter multiplies whole groups of 16 channels: the padded ones are executed too)
        nx_exec = (nx + 15) // 16 * 16 if kernel.startswith("k_gauss_gemm_filter") else nx
        flops = (2.0 if mfma else 3.0) * nx_exec * B * nd
        tflops = flops / (k1_us * 1e-6) / 1e12 if k1_us > 0 else 0.0
        alg_bytes = (8 * nx + 8) * B * nd                 # 1608 B per eval
        phys_bytes = 8 * nx * nd + 8 * nd + 8 * B * nx    # spectra once + thresholds + templates; flags
        traffic, traffic_source = pmc_traffic(kernel)
        res = {
            "metric": METRIC,
            "value": value, "unit": "likelihood evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "gensimple_%s %d spectra x %d channels per GPU, nlive %d: one nested-sampling iteration "
                                   "per step = prepare + K6 on a pool of %d live points + K3 on %d candidates + %d "
                                   "candidates scored and accept-tested against every spectrum + commit + advance, "
                                   "all on the device (BASELINE.json configs[%d])"
                                   % (args.workload, nd, nx, NLIVE, K, NCAND, B, 1 if args.workload == "horns" else 2),
                       "spectra_per_gpu": nd, "channels": nx, "candidates_per_step": B, "pool_points": K,
                       "parallelism": "datasets sharded x%d" % world,
                       "collectives": (None if not use_dist else
                                       "per step: pool all-gather, MAX all-reduce of the accept flags, fill-bit all-gather; "
                                       + ("RCCL called directly on the kernels' stream" if direct is not None
                                          else "torch.distributed (nccl)"))},
            "roofline": {"bound": "mfma" if mfma else "fp64_valu", "kernel": kernel, "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "launch_us": k1_us, "launches_timed": n_launch,
                         "flops_per_launch": flops,
                         "note": ("guarded accept filter on the matrix cores: the cross term of the expanded square as "
                                  "v_mfma_f64_16x16x4_f64 (fp64 matrix peak = fp64 vector peak on gfx950: 78.6 TFLOP/s; the "
                                  "instruction alone sustains 69.7, tools/probes/mfma_f64_probe.hip), decisions identical to "
                                  "the chain kernel's (k_exact_list re-scores what the error band cannot settle); flops = 2 x "
                                  "channels x candidates x spectra EXECUTED; in the reference's 3 flops per element the same "
                                  "launch is %.1f TFLOP/s" % (1.5 * tflops)) if mfma else
                                 ("B candidates are scored per pass over the spectra, so every spectrum byte is used B "
                                  "times and the kernel is bound by fp64 vector issue (v_add_f64 + v_fma_f64 per "
                                  "candidate, channel and spectrum: at most 0.75 of the FMA peak), not by HBM; "
                                  "flops = 3 x channels x candidates x spectra"),
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                 "frac_effective": (alg_bytes / (k1_us * 1e-6) / 1e9) / HBM_PEAK_GBS if k1_us > 0 else 0.0,
                                 "physical_bytes_per_launch": phys_bytes,
                                 "frac_physical": (phys_bytes / (k1_us * 1e-6) / 1e9) / HBM_PEAK_GBS if k1_us > 0 else 0.0,
                                 "note": "frac_effective = 1608 algorithmic bytes per eval / launch time / 8 TB/s (exceeds 1 "
                                         "because of the B-fold reuse: NOT a roofline fraction); frac_physical counts the "
                                         "bytes that have to move once"}},
        }
        res.update(other)
        if world == 1 and not args.no_e2e:
            res["e2e"] = e2e_leg(data, args.e2e_iterations)
            res["e2e_graph"] = e2e_leg(data, args.e2e_iterations, use_graph=True)
            if not args.no_e2e_full:
                # the whole analysis, not only its cheap first iterations (about 40 s)
                res["e2e_full"] = e2e_leg(data, 0, use_graph=True)
        if world == 1 and not args.no_hbm_leg:
            res["roofline_hbm_regime"] = hbm_regime_leg(lib, _lib)
            res["muse"] = muse_leg(lib, _lib)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_gauss(data, priortransform(np.random.RandomState(2).uniform(size=(64, NDIM))))
            if not args.no_e2e:
                # like for like: the first iterations of the SAME analysis on the CPU oracle's kernels and, with
                # the same cap, on the GPU (identical draws: the ratio of the two is the end-to-end speed-up)
                res["cpu_baseline"]["e2e"] = cpu_baseline_e2e(data, args.cpu_e2e_iterations)
                res["e2e_matched"] = e2e_leg(data, args.cpu_e2e_iterations, use_graph=True)
        print(json.dumps(res))

    lib.mdns_region_destroy(region)
    lib.mdns_joint_destroy(joint)
    lib.mdns_spectra_destroy(spectra)
    if use_dist:
        if direct is not None:
            direct.destroy()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------
def cpu_baseline_muse(y_t, v_t, templates, budget_s=15.0):
    """cmuselike on the host: the reference's OpenMP build (musefuse.py:505-508 loads it when
    OMP_NUM_THREADS > 1) on a bounded number of templates over the same spectra."""
    from oracle.oracle import Oracle, have_reference
    usable, info = host_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(usable))
    kind = "reference" if have_reference() else "port"
    orc = Oracle("reference-omp" if have_reference() else "port-omp")
    nd = y_t.shape[1]
    mask = np.ones(nd, dtype=np.bool_)
    out = np.zeros(nd)
    t0 = time.perf_counter()
    n = 0
    while True:
        orc.muse_like(y_t, v_t, templates[n % len(templates)], mask, Lout=out)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return {"value": n * nd / el, "unit": "likelihood evals/s", "cores": int(os.environ.get("OMP_NUM_THREADS", usable)),
            "kind": kind, "sample": "%d templates x %d spectra x %d channels, full mask, cmuselike OpenMP build, %.1f s"
                                    % (n, nd, y_t.shape[0], el),
            "binary": ref_provenance(), "host": info, "core_share": usable}, out.copy(), (n - 1) % len(templates)


def bench_muse(args):
    world, rank, use_dist, torch, dist, lib, _lib = setup_dist(args)
    from massivedatans_amd import gen
    nd, nx, B = args.muse_ndata, 4096, max(1, args.batch if args.batch_given else 1)
    cube = gen.muse_like(nd * world, nx)
    y_t = np.ascontiguousarray(cube["y"][:, rank * nd:(rank + 1) * nd])       # reference layout [nx, ndata]
    v_t = np.ascontiguousarray(cube["v"][:, rank * nd:(rank + 1) * nd])
    x = cube["x"]
    spectra = lib.mdns_spectra_create(_lib.ptr(x), _lib.ptr(y_t), _lib.ptr(v_t), nd, nx, 0)
    if not spectra:
        raise _lib.MdnsError(_lib.last_error())
    rng = np.random.RandomState(3)
    # template parameters (log_amp, z, log_width_scale, ratio1, ratio3) around the truth of the cube
    p5 = np.column_stack([rng.uniform(-0.3, 0.3, B), rng.uniform(0.0, 0.02, B), rng.uniform(-0.2, 0.2, B),
                          rng.uniform(0.5, 1.5, B), rng.uniform(0.5, 1.5, B)])
    templates = np.array([gen.muse_template(x, p) for p in p5])
    d_t = lib.mdns_dev_alloc(templates.nbytes)
    _lib.check(lib.mdns_h2d(d_t, _lib.ptr(templates), templates.nbytes), "h2d")
    d_L = lib.mdns_dev_alloc(B * nd * 8)

    def step():
        _lib.check(lib.mdns_muse_loglike_batch_dev(spectra, d_t, B, None, nd, d_L), "K2")

    dist_step = None
    if use_dist:
        # N ranks: one STEP = one chunk of a constrained draw over the sharded joint state (what
        # parallel.ShardedJointState does per chunk, BASELINE configs[4]): every rank scores the B
        # candidates against ITS spectra (templates on the device + K2 into the dense block + accept
        # test), the ranks MAX-reduce the B votes in place on the kernels' stream, every rank commits the
        # first candidate that has a vote, and the fill bits of all ranks are gathered -- between an
        # undo / prepare and an advance, so that every step is the same iteration.
        NLIVE_M = 100
        joint = lib.mdns_joint_create(spectra, NLIVE_M, 8)
        if not joint:
            raise _lib.MdnsError(_lib.last_error())
        live5 = np.column_stack([rng.uniform(-1.0, 1.0, NLIVE_M), rng.uniform(0.0, 0.02, NLIVE_M), rng.uniform(-0.5, 0.5, NLIVE_M),
                                 rng.uniform(0.2, 2.0, NLIVE_M), rng.uniform(0.2, 2.0, NLIVE_M)])
        _lib.check(lib.mdns_joint_init_muse3(joint, _lib.ptr(live5), None), "init")
        Lmin, arg, keep = np.empty(nd), np.empty(nd, dtype=np.int32), np.zeros((nd, lib.mdns_joint_keep_words(joint)), dtype=np.uint64)
        _lib.check(lib.mdns_joint_prepare(joint, _lib.ptr(Lmin), _lib.ptr(arg), _lib.ptr(keep)), "prepare")
        votes = lib.mdns_joint_votes_dev
        nbits = (nd + 63) // 64
        bits = np.zeros(nbits + 1, dtype=np.uint64)
        t_allbits = torch.zeros(world * nbits, dtype=torch.int64, device="cuda")
        accepted = C.c_int(-1)
        direct, rccl, stream = COMM["direct"], COMM["rccl"], COMM["stream"]
        from massivedatans_amd.parallel import device_view
        state = {"advanced": False}

        def dist_step():
            if state["advanced"]:
                _lib.check(lib.mdns_joint_undo_advance_dev(joint), "undo advance")
            _lib.check(lib.mdns_joint_prepare_dev(joint), "prepare")
            _lib.check(lib.mdns_backend_draw_begin(joint, None, nd), "draw_begin")
            _lib.check(lib.mdns_backend_draw_score(joint, _lib.ptr(p5), B, None), "templates + K2 + accept")
            d_votes = votes(joint)
            if direct is not None:
                direct.all_reduce(d_votes, d_votes, B, rccl.INT32, rccl.MAX, stream)
            else:
                dist.all_reduce(device_view(d_votes, (B,), "<i4"), op=dist.ReduceOp.MAX)
            _lib.check(lib.mdns_backend_draw_commit(joint, C.addressof(accepted), _lib.ptr(bits)), "commit")
            t_bits = torch.from_numpy(bits[:nbits].view(np.int64)).cuda()
            if direct is not None:
                direct.all_gather(t_bits.data_ptr(), t_allbits.data_ptr(), nbits, rccl.INT64, stream)
            else:
                dist.all_gather_into_tensor(t_allbits, t_bits)
            _lib.check(lib.mdns_joint_advance_dev(joint), "advance")
            state["advanced"] = True

    def fence():
        _lib.check(lib.mdns_sync(), "sync")
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if dist_step is not None:
        step = dist_step
    lib.mdns_profile_every(1)
    lib.mdns_profile(2)
    elapsed = timed(step, fence, args, use_dist, torch, dist, lib, _lib)
    n_launch, k2_us = read_profile(lib, 1)
    lib.mdns_profile(0)
    kernel = (lib.mdns_profile_kernel(1) or b"").decode()
    L = np.empty(B * nd)
    if dist_step is None:
        _lib.check(lib.mdns_d2h(_lib.ptr(L), d_L, B * nd * 8), "d2h")
    L = L.reshape(B, nd)

    if rank == 0:
        value = B * nd * world * args.steps / elapsed
        alg_bytes = (16 * nx + 8) * B * nd                # 65 544 B per eval
        phys_bytes = 16 * nx * nd + 8 * B * nd + 8 * B * nx
        # per candidate, channel and spectrum: y w m, m m w, then (y - s m)^2 w = 4 multiplies + 3 fused
        flops = 10.0 * nx * B * nd
        gbs = phys_bytes / (k2_us * 1e-6) / 1e9 if k2_us > 0 else 0.0
        tflops = flops / (k2_us * 1e-6) / 1e12 if k2_us > 0 else 0.0
        hbm_bound = B < 8
        traffic, traffic_source = pmc_traffic(kernel)
        res = {
            "metric": METRIC, "value": value, "unit": "likelihood evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "MUSE-style cmuselike: %d spectra x %d channels with per-pixel variances per GPU (one GPU's "
                                   "share of BASELINE.json configs[4]: 50 000 over 8), %d template%s per pass; y and 1/v = "
                                   "%.0f MB streamed from HBM" % (nd, nx, B, "" if B == 1 else "s", 16 * nx * nd / 1e6),
                       "spectra_per_gpu": nd, "channels": nx, "candidates_per_step": B,
                       "step": ("one chunk of a constrained draw over the sharded joint state: templates + K2 + accept test per "
                                "rank, MAX all-reduce of the votes, commit, all-gather of the fill bits" if dist_step is not None
                                else "the K2 kernel alone"),
                       "collectives": ("none" if dist_step is None else "RCCL called directly on the kernels' stream"
                                       if COMM["direct"] is not None else "torch.distributed (nccl)"),
                       "parallelism": "datasets sharded x%d" % world},
            "roofline": ({"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                          "launch_us": k2_us, "launches_timed": n_launch, "bytes_per_launch": phys_bytes,
                          "algorithmic_bytes_per_launch": alg_bytes,
                          "note": "one fused pass over y and 1/v (the reference makes two strided passes): 65 544 B per eval "
                                  "at B = 1; with B templates the rows are read once, bytes_per_launch is what moves"}
                         if hbm_bound else
                         {"bound": "fp64_valu", "kernel": kernel, "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS, "traffic": traffic,
                          "traffic_source": traffic_source, "launch_us": k2_us, "launches_timed": n_launch,
                          "flops_per_launch": flops, "hbm_frac_physical": gbs / HBM_PEAK_GBS,
                          "note": "10 flops per (template, channel, spectrum): two dot products and the residual sum"}),
        }
        if world == 1 and not args.no_e2e:
            res["e2e"] = e2e_muse_leg(x, y_t, v_t, args.e2e_iterations, jitter=True)
            res["e2e_no_noise"] = e2e_muse_leg(x, y_t, v_t, args.e2e_iterations, jitter=False)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"], ref_L, which = cpu_baseline_muse(y_t, v_t, templates)
            err = np.max(np.abs(L[which] - ref_L) / np.abs(ref_L))
            assert err < 1e-9, ("K2 bench output differs from the CPU path", err)
            res["parity_vs_cpu_max_rel"] = float(err)
        print(json.dumps(res))
    lib.mdns_spectra_destroy(spectra)
    if use_dist:
        if COMM["direct"] is not None:
            COMM["direct"].destroy()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="candidates scored per step (B): default 256 (horns/nothing), 1 (muse)")
    ap.add_argument("--ndata", type=int, default=10000, help="spectra per GPU (horns / nothing)")
    ap.add_argument("--muse-ndata", type=int, default=6250, help="spectra per GPU (muse)")
    ap.add_argument("--pool", type=int, default=4 * NLIVE, help="unique live points in the pool (K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the 1.6 GB one-pass HBM-regime measurement")
    ap.add_argument("--no-e2e", action="store_true", help="skip the capped complete analysis")
    ap.add_argument("--e2e-iterations", type=int, default=400)
    ap.add_argument("--cpu-e2e-iterations", type=int, default=100, help="cap of the CPU-oracle run of the same analysis (cpu_baseline.e2e)")
    ap.add_argument("--no-e2e-full", action="store_true", help="skip the complete analysis (about 40 s)")
    ap.add_argument("--event-every", type=int, default=4, help="time every n-th launch of the dominant kernel")
    ap.add_argument("--no-events", action="store_true", help="no per-launch events in the timed loop (roofline empty)")
    ap.add_argument("--workload", default="horns", choices=["horns", "nothing", "muse"])
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    args.batch_given = args.batch is not None
    if args.batch is None:
        args.batch = 256
    if args.workload == "muse":
        bench_muse(args)
    else:
        bench_gauss(args)


if __name__ == "__main__":
    main()
