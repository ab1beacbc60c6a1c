#!/usr/bin/env python3
"""Kernel-level micro-benchmarks (SURVEY.md 8(d) "kernel-level bench"): device time per call of
each hot-path entry point over sweeps of batch size, mask density and pool size, measured with
HIP events on the launch stream around back-to-back calls.  One JSON line per case.

    python bench_kernels.py [k1] [k1big] [k2] [k3] [k6]      (default: k1 k2 k3 k6)
"""
import ctypes as C
import json
import sys

import numpy as np

from massivedatans_amd import _lib, gen

REPS = 30


def timed(lib, which, fn, reps=REPS, warm=3):
    """Microseconds per CALL: `reps` calls back to back between two events on the launch stream,
    so every kernel of the call counts (template kernel, compaction of a selection, ...)."""
    for _ in range(warm):
        fn()
    lib.mdns_sync()
    e0, e1 = lib.mdns_event_create(), lib.mdns_event_create()
    lib.mdns_event_record(e0)
    for _ in range(reps):
        fn()
    lib.mdns_event_record(e1)
    ms = lib.mdns_event_elapsed_ms(e0, e1)
    lib.mdns_event_destroy(e0)
    lib.mdns_event_destroy(e1)
    return 1e3 * ms / reps


def dev(lib, a):
    a = np.ascontiguousarray(a)
    p = lib.mdns_dev_alloc(a.nbytes)
    _lib.check(lib.mdns_h2d(p, _lib.ptr(a), a.nbytes), "h2d")
    return p


def k1(lib, ndata=10000):
    d = gen.horns(ndata)
    nx = d["y"].shape[0]
    sp = lib.mdns_spectra_create(_lib.ptr(d["x"]), _lib.ptr(d["y"]), None, ndata, nx, 0)
    rng = np.random.RandomState(1)
    # masks of SURVEY 8(d): 100 %, (50 %,) 10 %, 1 % of the data sets and a single one
    for frac in (1.0, 0.5, 0.1, 0.01, "single"):
        if frac == "single":
            rows = np.array([ndata // 2], dtype=np.int32)
            d_rows, M = dev(lib, rows), 1
        elif frac < 1.0:
            rows = np.flatnonzero(rng.uniform(size=ndata) < frac).astype(np.int32)
            d_rows, M = dev(lib, rows), len(rows)
        else:
            d_rows, M = None, ndata
        for B in (1, 2, 4, 8, 16, 64, 256, 1024, 4096):
            cube = rng.uniform(size=(B, 3))
            params = np.column_stack([10 ** (cube[:, 0] * 2 - 2), cube[:, 1] * 400 + 400, 10 ** (cube[:, 2] * 2)])
            d_p = dev(lib, params)
            d_L = lib.mdns_dev_alloc(B * M * 8)
            us = timed(lib, 0, lambda: lib.mdns_gauss_loglike_batch_dev(sp, d_p, B, 0.01, d_rows, M, d_L))
            evals = B * M
            print(json.dumps({"kernel": "K1", "ndata": ndata, "nx": nx, "mask": frac, "M": M, "B": B, "us": us,
                              "evals_per_s": evals / (us * 1e-6),
                              "alg_GBps": evals * (8 * nx + 8) / (us * 1e-6) / 1e9,
                              "phys_GBps": (M * nx * 8 + evals * 8) / (us * 1e-6) / 1e9,
                              "fp64_valu_frac": 3.0 * nx * evals / (us * 1e-6) / 78.6e12}), flush=True)
            lib.mdns_dev_free(d_p)
            lib.mdns_dev_free(d_L)
    lib.mdns_spectra_destroy(sp)


def k1big(lib, ndata=1000000):
    """K1 on a spectra set far larger than the 256 MiB Infinity Cache (1.6 GB): the
    HBM-bound regime of the row kernel (one pass, B = 1..4) and of the lane kernel."""
    nx = 200
    rng = np.random.RandomState(ndata)
    x = gen.wavelength_grid()
    y = np.ascontiguousarray(rng.normal(0, 0.01, size=(ndata, nx)))      # [ndata, nx]: no transpose needed
    sp = lib.mdns_spectra_create(_lib.ptr(x), _lib.ptr(y), None, ndata, nx, 1)
    if not sp:
        raise _lib.MdnsError(_lib.last_error())
    for B in (1, 2, 4, 16, 64, 256):
        cube = rng.uniform(size=(B, 3))
        params = np.column_stack([10 ** (cube[:, 0] * 2 - 2), cube[:, 1] * 400 + 400, 10 ** (cube[:, 2] * 2)])
        d_p = dev(lib, params)
        d_L = lib.mdns_dev_alloc(B * ndata * 8)
        us = timed(lib, 0, lambda: lib.mdns_gauss_loglike_batch_dev(sp, d_p, B, 0.01, None, ndata, d_L), reps=10)
        evals = B * ndata
        print(json.dumps({"kernel": "K1", "ndata": ndata, "nx": nx, "mask": 1.0, "M": ndata, "B": B, "us": us,
                          "evals_per_s": evals / (us * 1e-6),
                          "alg_GBps": evals * (8 * nx + 8) / (us * 1e-6) / 1e9,
                          "phys_GBps": (ndata * nx * 8 + evals * 8) / (us * 1e-6) / 1e9,
                          "fp64_valu_frac": 3.0 * nx * evals / (us * 1e-6) / 78.6e12}), flush=True)
        lib.mdns_dev_free(d_p)
        lib.mdns_dev_free(d_L)
    lib.mdns_spectra_destroy(sp)


def k2(lib, ndata=4096, nx=4096):
    cube = gen.muse_like(ndata, nx=nx)
    sp = lib.mdns_spectra_create(_lib.ptr(cube["x"]), _lib.ptr(cube["y"]), _lib.ptr(cube["v"]), ndata, nx, 0)
    rng = np.random.RandomState(2)
    for frac in (1.0, 0.1, 0.01, "single"):
        if frac == "single":
            rows = np.array([ndata // 2], dtype=np.int32)
        elif frac < 1.0:
            rows = np.flatnonzero(rng.uniform(size=ndata) < frac).astype(np.int32)
        else:
            rows = None
        d_rows, M = (dev(lib, rows), len(rows)) if rows is not None else (None, ndata)
        for B in (1, 4, 16, 64):
            pars = np.column_stack([rng.uniform(-0.5, 0.5, B), rng.uniform(0, 0.02, B), rng.uniform(-0.1, 0.2, B),
                                    rng.uniform(0.5, 1.5, B), rng.uniform(0.5, 1.5, B)])
            d_p = dev(lib, pars)
            d_L = lib.mdns_dev_alloc(B * M * 8)
            us = timed(lib, 1, lambda: lib.mdns_muse3_loglike_batch_dev(sp, d_p, B, d_rows, M, d_L), reps=10)
            evals = B * M
            print(json.dumps({"kernel": "K2", "ndata": ndata, "nx": nx, "mask": frac, "M": M, "B": B, "us": us,
                              "evals_per_s": evals / (us * 1e-6),
                              "alg_GBps": evals * (16 * nx + 8) / (us * 1e-6) / 1e9,
                              "phys_GBps": (M * nx * 16 + evals * 8) / (us * 1e-6) / 1e9}), flush=True)
            lib.mdns_dev_free(d_p)
            lib.mdns_dev_free(d_L)
    lib.mdns_spectra_destroy(sp)


def k3k6(lib, which):
    rng = np.random.RandomState(3)
    for ndim in (3, 5):
        for K in (100, 400, 1000, 10000, 100000):
            pool = rng.uniform(size=(K, ndim))
            d_pool = dev(lib, pool)
            if "k3" in which:
                for M in (1000, 10000):
                    cands = rng.uniform(size=(M, ndim))
                    d_c, d_n = dev(lib, cands), lib.mdns_dev_alloc(M * 4)
                    us = timed(lib, 2, lambda: lib.mdns_count_within_dev(d_pool, K, ndim, 0.1, d_c, M, d_n))
                    print(json.dumps({"kernel": "K3", "ndim": ndim, "K": K, "M": M, "us": us,
                                      "pairs_per_s": K * M / (us * 1e-6)}), flush=True)
                    lib.mdns_dev_free(d_c)
                    lib.mdns_dev_free(d_n)
            if "k6" in which and K <= 10000 * (10 if ndim == 3 else 1):
                chosen = np.zeros((K, 10))
                for b in range(10):
                    chosen[rng.choice(np.arange(K), size=K, replace=True), b] = 1.
                d_ch, d_r = dev(lib, chosen), lib.mdns_dev_alloc(80)
                us = timed(lib, 3, lambda: lib.mdns_bootstrap_round_maxsq_dev(d_pool, K, ndim, d_ch, 10, d_r),
                           reps=10 if K >= 100000 else REPS)
                print(json.dumps({"kernel": "K6", "ndim": ndim, "K": K, "B": 10, "us": us,
                                  "pairs_per_s": K * K / (us * 1e-6)}), flush=True)
                lib.mdns_dev_free(d_ch)
                lib.mdns_dev_free(d_r)
            lib.mdns_dev_free(d_pool)


def main():
    which = [a.lower() for a in sys.argv[1:]] or ["k1", "k2", "k3", "k6"]
    lib = _lib.require_device()
    if "k1" in which:
        k1(lib)
    if "k1big" in which:
        k1big(lib)
    if "k2" in which:
        k2(lib)
    if "k3" in which or "k6" in which:
        k3k6(lib, which)


if __name__ == "__main__":
    main()
