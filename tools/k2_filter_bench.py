#!/usr/bin/env python3
"""The band test of a MUSE chunk (mdns_backend_draw_band) through the matrix-core filter against the exact
row kernels: same decisions, and the time of a chunk either way.

    python tools/k2_filter_bench.py [ndata] [nx] [B] [reps]

Thresholds are planted on the candidates' own likelihoods (at relative distances from 1e-3 down to 0 on
both sides), so that the filter has pairs it cannot settle and must hand to the exact kernels."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import _lib, gen, jointstate, musefuse
from massivedatans_amd.like import MuseSpectra


def planted_state(ndata, nx, nlive, B, seed, offsets, rows=None):
    """spectra, a joint state whose thresholds sit next to the likelihoods of `params`, and those"""
    rng = np.random.RandomState(seed)
    data = gen.muse_like(ndata, nx)
    sp = MuseSpectra(data["x"], data["y"], data["v"])
    st = jointstate.MuseJointState(sp, nlive, shelf_cap=4)
    st.init(musefuse.priortransform_batch(rng.uniform(size=(nlive, 5))))
    params = musefuse.priortransform_batch(rng.uniform(size=(B, 5)))
    L = sp.loglike_batch_lines(params, None)                       # exact kernels, [B, ndata]
    live = st.live_matrix()
    # data set d: threshold = L of candidate (d mod B) moved by one of the offsets; the other live points far above
    which = np.arange(ndata) % B
    off = np.asarray(offsets)[rng.randint(len(offsets), size=ndata)]
    base = L[which, np.arange(ndata)]
    thr = base + off * np.abs(base)
    live[:] = np.maximum(live, thr[None, :] + 1e6)
    live[0] = thr
    lib = st._lib
    st._check(lib.mdns_joint_set_live(st._h, _lib.ptr(np.ascontiguousarray(live))), "mdns_joint_set_live")
    st.prepare()
    return sp, st, params, L, thr


def band(st, params, rows, bound, mode):
    lib = st._lib
    lib.mdns_muse_filter_mode(mode)
    B = len(params)
    M = st.ndata if rows is None else len(rows)
    st._check(lib.mdns_backend_draw_begin(st._h, _lib.ptr(rows) if rows is not None else None, M), "draw_begin")
    cap = 4096
    status = np.zeros(B, dtype=np.int32)
    npairs = C.c_int(0)
    pb, pk = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    pL, pthr = np.zeros(cap), np.zeros(cap)
    p = np.ascontiguousarray(params)
    t0 = time.perf_counter()
    st._check(lib.mdns_backend_draw_band(st._h, _lib.ptr(p), B, _lib.ptr(bound), _lib.ptr(status), C.byref(npairs), _lib.ptr(pb), _lib.ptr(pk),
                                         _lib.ptr(pL), _lib.ptr(pthr), cap), "draw_band")
    dt = time.perf_counter() - t0
    n = min(npairs.value, cap)
    order = np.lexsort((pk[:n], pb[:n]))
    return status, npairs.value, pb[:n][order], pk[:n][order], pL[:n][order], pthr[:n][order], dt


def stats(lib):
    out = (C.c_longlong * 4)()
    lib.mdns_muse_filter_stats(out)
    return list(out)


if __name__ == "__main__":
    ndata = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
    nx = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    out = {"ndata": ndata, "nx": nx, "B": B}
    # decisions: planted thresholds, no noise bound
    sp, st, params, L, thr = planted_state(ndata, nx, 8, B, 1, [-1e-3, -1e-9, -1e-13, 0.0, 1e-13, 1e-9, 1e-3])
    lib = st._lib
    zero = np.zeros(B)
    s0 = stats(lib)
    a = band(st, params, None, zero, 0)
    f = band(st, params, None, zero, 1)
    s1 = stats(lib)
    out["planted"] = {"status_equal": bool(np.array_equal(a[0], f[0])), "npairs_exact": a[1], "npairs_after_filter": f[1],
                      "pairs_equal": bool(a[1] == f[1] and all(np.array_equal(x, y) for x, y in zip(a[2:6], f[2:6]))),
                      "filter_chunks": s1[0] - s0[0], "rescored": s1[1] - s0[1]}
    st.close(); sp.close()
    # time: thresholds far from every candidate (nothing to list), a noise bound as in a run
    sp, st, params, L, thr = planted_state(ndata, nx, 8, B, 2, [-1e-2, 1e-2])
    lib = st._lib
    bound = np.full(B, 5e-5)
    for mode, name in ((0, "exact"), (1, "filter")):
        band(st, params, None, bound, mode)
        ts = []
        for _ in range(reps):
            ts.append(band(st, params, None, bound, mode)[-1])
        out[name + "_us"] = 1e6 * float(np.median(ts))
        out[name + "_kernel"] = (lib.mdns_profile_kernel(1) or b"").decode()
    a, f = band(st, params, None, bound, 0), band(st, params, None, bound, 1)
    out["status_equal"] = bool(np.array_equal(a[0], f[0]))
    out["npairs"] = [a[1], f[1]]
    out["stats"] = stats(lib)
    flops = 4.0 * ndata * ((nx + 15) // 16 * 16) * B
    out["filter_tflops_incl_round_trip"] = flops / (out["filter_us"] * 1e-6) / 1e12
    print(json.dumps(out))
