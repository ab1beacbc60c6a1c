#!/usr/bin/env python3
"""Constrained-draw likelihood throughput on MI355X (BASELINE.json metric).

One STEP = one constrained-draw pass of the hot path over one batch of synthetic input, with
everything resident in HBM when the timed region starts:

  1. (N > 1 only) RCCL all-gather of the shared live-point pool, K/N points per rank
  2. K6  RadFriends safe radius of the pool (10 bootstrap rounds)      -> 10 doubles to the host
  3. K3  membership count of 1000 box candidates against the pool
  4. K1  B candidate lines x every spectrum of this rank's shard        -> L[B, ndata]

Workload at N = 1: config C2 of BASELINE.json, gensimple_horns 10 000 spectra x 200 channels,
nlive 100 (pool of 4*nlive unique live points).  N > 1: weak scaling, 10 000 spectra per GPU
cut from horns(10 000 * N), one process per GPU, no data-path collective besides (1).

`value` = likelihood evaluations (candidate, spectrum pairs) per second over all ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (spec sheet; = half the fp32 vector rate)
NDIM = 3
NLIVE = 100
NBOOT = 10                     # clustering/radfriendsregion.py:59
NCAND = 1000                   # clustering/radfriendsregion.py:124


def priortransform(cube):
    """sample.py:52-58 + sig = 10**log_sig (sample.py:103), vectorised over rows."""
    return np.column_stack([10 ** (cube[:, 0] * 2 - 2), cube[:, 1] * 400 + 400, 10 ** (cube[:, 2] * 2)])


def cpu_baseline(data, params, budget_s=12.0):
    """Reference CPU path for K1 on the host cores of this box, bounded sample.
    Uses the compiled reference (oracle/_ref) when it travelled with the snapshot, else our
    C restatement.  Serial, like the reference (sample.py:81 never loads clike-parallel)."""
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
    res = {"value": n * nd / el, "unit": "likelihood evals/s", "cores": 1, "kind": kind,
           "sample": "%d candidates x %d spectra x 200 channels, full mask, serial clike loop, %.1f s" % (n, nd, el)}
    # best-effort multi-threaded form (dataset-parallel OpenMP restatement), for context only
    try:
        # the GPU box gives this job a 16-core share however many cores the host reports
        ncores = min(os.cpu_count() or 1, 16)
        os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
        omp = Oracle("port-omp")
        t0 = time.perf_counter()
        m = 0
        while time.perf_counter() - t0 < 4.0:
            p = params[m % len(params)]
            out[:] = 0
            omp.gauss_like(x, y, p[0], p[1], p[2], 0.01, mask, Lout=out)
            m += 1
        res["value_openmp"] = m * nd / (time.perf_counter() - t0)
        res["cores_openmp"] = int(os.environ.get("OMP_NUM_THREADS", ncores))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="candidates scored per step (B)")
    ap.add_argument("--ndata", type=int, default=10000, help="spectra per GPU")
    ap.add_argument("--pool", type=int, default=4 * NLIVE, help="unique live points in the pool (K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the 1.6 GB one-pass HBM-regime measurement")
    ap.add_argument("--event-every", type=int, default=4, help="time every n-th launch of the dominant kernel")
    ap.add_argument("--no-events", action="store_true", help="no per-launch events in the timed loop (roofline empty)")
    ap.add_argument("--workload", default="horns", choices=["horns", "nothing"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    dist = None
    torch = None
    # MDNS_BENCH_FORCE_DIST=1 exercises the collective path with a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("MDNS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        # torch first: it brings its own copy of the HIP runtime (same soname); loaded first it
        # becomes the one runtime of the process, which libmdns_hip.so then binds to as well
        import torch
        import torch.distributed as dist
        # one visible device per rank (HIP_VISIBLE_DEVICES set by a launcher) or all of them
        device_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(device_index)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
    else:
        device_index = local_rank

    from massivedatans_amd import _lib, gen
    os.environ.setdefault("MDNS_DEVICE", str(device_index))
    lib = _lib.require_device()
    if use_dist:
        # The kernels go on torch's CURRENT stream, which the (synchronous-API) RCCL collective
        # orders itself against on both sides: K6 of a step cannot start before that step's
        # all-gather has delivered the pool.  A stream of our own, because torch's default
        # stream is the null stream (handle 0), which mdns_set_stream reads as "library stream".
        bench_stream = torch.cuda.Stream()
        torch.cuda.set_stream(bench_stream)
        assert bench_stream.cuda_stream != 0
        _lib.check(lib.mdns_set_stream(C.c_void_p(bench_stream.cuda_stream)), "mdns_set_stream")

    # ---- synthetic input, resident before the timed region --------------------------------
    nd, B, K = args.ndata, args.batch, args.pool
    make = gen.horns if args.workload == "horns" else gen.nothing
    full = make(nd * world)
    shard = np.ascontiguousarray(full["y"][:, rank * nd:(rank + 1) * nd])
    data = {"x": full["x"], "y": shard}
    spectra = lib.mdns_spectra_create(_lib.ptr(data["x"]), _lib.ptr(shard), None, nd, shard.shape[0], 0)
    if not spectra:
        raise _lib.MdnsError(_lib.last_error())
    nx = shard.shape[0]

    rng = np.random.RandomState(1)                      # sample.py:162
    params = priortransform(rng.uniform(size=(B, NDIM)))
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
    d_L = lib.mdns_dev_alloc(B * nd * 8)
    d_counts = lib.mdns_dev_alloc(NCAND * 4)
    if use_dist:
        assert K % world == 0, "pool size must divide over the ranks"
        t_pool = torch.empty((K, NDIM), dtype=torch.float64, device="cuda")
        t_mine = torch.from_numpy(pool[rank * (K // world):(rank + 1) * (K // world)].copy()).cuda()
        d_pool = t_pool.data_ptr()
    else:
        d_pool = dev(pool)
    region = lib.mdns_region_wrap_dev(d_pool, K, NDIM)     # the pool buffer is refilled in place
    if not region:
        raise _lib.MdnsError(_lib.last_error())

    def step():
        if use_dist:
            dist.all_gather_into_tensor(t_pool, t_mine)
        # K6, then radius + membership threshold finished on the device: K3 follows in stream order
        _lib.check(lib.mdns_region_bootstrap_radius_async(region, d_chosen, NBOOT), "K6")
        _lib.check(lib.mdns_region_count_dev(region, d_cands, NCAND, d_counts), "K3")
        _lib.check(lib.mdns_gauss_loglike_batch_dev(spectra, d_params, B, 0.01, None, nd, d_L), "K1")
        # the host needs the radius (bounding box of the next proposals): fetched while K1 runs
        radius = lib.mdns_region_radius(region)
        if radius != radius:
            raise _lib.MdnsError(_lib.last_error())

    def fence():
        _lib.check(lib.mdns_sync(), "sync")
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # events only around the dominant kernel (every timed launch adds two event records), and its
    # launches are sampled (every 4th), which keeps the cost of measuring out of `value`
    lib.mdns_profile_every(args.event_every)
    lib.mdns_profile(1 if not args.no_events else 0)
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

    # per-launch duration of the dominant kernel, HIP events on the launch stream
    n_launch, tot_ms = C.c_longlong(0), C.c_double(0)
    lib.mdns_profile_read(0, C.byref(n_launch), C.byref(tot_ms))
    k1_ms = tot_ms.value / max(1, n_launch.value)
    kernel = (lib.mdns_profile_kernel(0) or b"").decode()      # e.g. "k_gauss_cols<8, 1>"
    # the geometry kernels of the step, timed in a few extra (unreported) steps
    lib.mdns_profile_every(1)
    lib.mdns_profile(4 | 8)
    for _ in range(min(args.steps, 50)):
        step()
    other = {}
    for which, name in ((2, "count_within"), (3, "bootstrap")):
        n2, ms2 = C.c_longlong(0), C.c_double(0)
        lib.mdns_profile_read(which, C.byref(n2), C.byref(ms2))
        other[name + "_us"] = 1e3 * ms2.value / max(1, n2.value)
    lib.mdns_profile(0)
    fence()

    # sanity: the timed launches produced the right numbers (first candidate, a few spectra)
    L = np.empty(B * nd)
    _lib.check(lib.mdns_d2h(_lib.ptr(L), d_L, B * nd * 8), "d2h")
    L = L.reshape(B, nd)
    ypred = params[0, 0] * np.exp(-0.5 * ((params[0, 1] - data["x"]) / params[0, 2]) ** 2)
    want = -0.5 * (((ypred.reshape((-1, 1)) - shard[:, :64]) / 0.01) ** 2).sum(axis=0)
    assert np.allclose(L[0, :64], want, rtol=1e-10), "bench output check failed"

    if rank == 0:
        evals_per_step = B * nd * world
        value = evals_per_step * args.steps / elapsed
        bytes_per_eval = 8 * nx + 8                       # y row read + L written
        alg_bytes = bytes_per_eval * B * nd               # per launch (one rank)
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        phys_bytes = 8 * nx * nd + 8 * B * nd + 8 * B * 512   # spectra once + L + templates
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:       # HBM bytes per launch of that kernel, from the committed --pmc passes
                traffic = json.load(open(pmc))["kernels"]["mdns::" + kernel]["hbm_bytes"]
            except Exception:      # noqa: BLE001
                traffic = None
        res = {
            "metric": "likelihood evals/sec across N datasets per constrained draw",
            "value": value, "unit": "likelihood evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "gensimple_%s %d spectra x %d channels per GPU, nlive %d, pool %d live points, "
                                   "%d candidates per draw step (BASELINE.json configs[1])"
                                   % (args.workload, nd, nx, NLIVE, K, B),
                       "spectra_per_gpu": nd, "channels": nx, "candidates_per_step": B, "pool_points": K,
                       "parallelism": "datasets sharded x%d" % world},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_us": 1e3 * k1_ms, "launches_timed": int(n_launch.value),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "physical_bytes_per_launch": phys_bytes,
                         "frac_physical": (phys_bytes / (k1_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if k1_ms > 0 else 0.0,
                         "fp64_valu_frac": (3.0 * nx * B * nd / (k1_ms * 1e-3) / 1e12) / FP64_VALU_PEAK_TFLOPS if k1_ms > 0 else 0.0,
                         "note": "algorithmic bytes = 1608 B per (candidate, spectrum) eval; with B candidates scored per "
                                 "pass each spectrum is read once and reused B times, so frac can exceed 1 "
                                 "(effective, not physical, bandwidth); frac_physical counts bytes actually moved"},
        }
        res.update(other)
        if not args.no_hbm_leg and world == 1:
            res["roofline_hbm_regime"] = hbm_regime_leg(lib, _lib)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(data, params)
        print(json.dumps(res))

    lib.mdns_region_destroy(region)
    lib.mdns_spectra_destroy(spectra)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
