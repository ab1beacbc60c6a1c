#!/usr/bin/env python3
"""Times the device grouping (csrc/mdns_groups.hip) on id matrices shaped like those of a C2 run:
python tools/groups_bench.py [ndata nlive ndistinct nselected]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd.grouping import DeviceGroups

ndata, nlive, ndistinct, nsel = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (10000, 100, 9000, 2000)))
rng = np.random.RandomState(1)
# a core of ids most data sets share plus private tails, like late iterations of a run
lp = np.empty((nlive, ndata), dtype=np.int32)
for d in range(ndata):
    lp[:, d] = rng.choice(ndistinct, size=nlive, replace=False)
for shape in ("random", "shared-core"):
    if shape == "shared-core":
        lp[: nlive // 2, :] = np.arange(nlive // 2)[:, None]
    dg = DeviceGroups(lp)
    npoints = 150000
    for rows in (None, np.sort(rng.choice(ndata, size=nsel, replace=False))):
        dg.components(rows, npoints)
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            ncomp, ids = dg.components(rows, npoints)
        dt = (time.perf_counter() - t0) / n
        print("%s, %s data sets: %d components, %d ids, %.1f us per call, %.1f rounds" % (shape, "all" if rows is None else len(rows), ncomp, len(ids), dt * 1e6, dg.mean_rounds()))
    dg.close()
