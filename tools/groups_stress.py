#!/usr/bin/env python3
"""Stress of the device grouping (csrc/mdns_groups.hip): long chains (many rounds), many random
bipartite graphs, repeated calls -- every result against scipy.  python tools/groups_stress.py [trials]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from massivedatans_amd.grouping import DeviceGroups
from test_groups import cpu_groups, clustered_ids

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(123)
bad = 0
# 1. paths: data set d holds ids d and d + 1 -- the diameter is the number of data sets
for n in (10, 100, 1000, 5000, 20000):
    lp = np.vstack([np.arange(n), np.arange(n) + 1])
    order = rng.permutation(n)                       # and with the data sets shuffled along the path
    for name, mat in (("path", lp), ("shuffled path", lp[:, order])):
        dg = DeviceGroups(mat)
        t0 = time.perf_counter()
        ncomp, ids = dg.components(None, n + 1)
        dt = time.perf_counter() - t0
        ok = ncomp == 1 and np.array_equal(ids, np.arange(n + 1))
        bad += not ok
        print("%s of %d data sets: %d component(s), %.1f rounds, %.2f ms%s" % (name, n, ncomp, dg.mean_rounds(), dt * 1e3, "" if ok else "  WRONG"))
        dg.close()
# 2. many random graphs, each asked several times with different selections
t0 = time.time()
for trial in range(trials):
    nlive = int(rng.choice([1, 2, 3, 8, 40, 100]))
    ndata = int(rng.choice([2, 17, 300, 2500, 9000]))
    ncl = int(rng.choice([1, 2, 5, 60]))
    lp = clustered_ids(rng, nlive, ndata, ncl, max(2 * nlive, int(rng.choice([nlive + 1, 50, 400]))))
    npoints = int(lp.max()) + 1
    dg = DeviceGroups(lp)
    for rep in range(4):
        rows = None if rep == 0 else np.flatnonzero(rng.uniform(size=ndata) < rng.choice([0.05, 0.5, 0.9]))
        if rows is not None and len(rows) == 0:
            continue
        want = cpu_groups(lp, np.arange(ndata) if rows is None else rows)
        got = dg.groups(rows, npoints)
        ok = len(got) == len(want) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(got, want))
        if not ok:
            bad += 1
            print("MISMATCH trial %d rep %d: nlive %d ndata %d clusters %d: %d vs %d groups" % (trial, rep, nlive, ndata, ncl, len(got), len(want)))
    dg.close()
print("%d random graphs x 4 selections in %.0f s, %d wrong" % (trials, time.time() - t0, bad))
sys.exit(1 if bad else 0)
