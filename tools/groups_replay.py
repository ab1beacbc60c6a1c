#!/usr/bin/env python3
"""Replays the groupings of one iteration of a real run off line (no GPU): the id matrix and the
selections of its passes as dumped by MDNS_CORE_DUMP=<iteration>:<file> (csrc/host_sampler.cpp), through
the sampler core's debug entry points -- fresh components against the ones kept up to date.
    python tools/groups_replay.py gpurun_out/r04_dump/iter650.bin.gz [repeats]"""
import ctypes as C, gzip, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_core as t
from massivedatans_amd import core

raw = (gzip.open if sys.argv[1].endswith(".gz") else open)(sys.argv[1], "rb").read()
nrun, nlive, npile = np.frombuffer(raw[:24], dtype=np.int64)
at = 24
lp = np.frombuffer(raw[at:at + 4 * nrun * nlive], dtype=np.int32).reshape(nrun, nlive)
at += 4 * nrun * nlive
passes = []
while at < len(raw):
    M, focussed = np.frombuffer(raw[at:at + 16], dtype=np.int64)
    at += 16
    passes.append((bool(focussed), np.frombuffer(raw[at:at + 4 * M], dtype=np.int32).copy()))
    at += 4 * M
print("%d running data sets x %d live points, %d points; %d passes: focussed selections of %s data sets"
      % (nrun, nlive, npile, len(passes), [len(s) for f, s in passes if f][:12] + ["..."]))
L, h, keep = t._core(int(nlive), int(nrun))
ids = np.ascontiguousarray(lp.T)
assert L.mdns_core_debug_set_ids(h, ids.ctypes.data, int(npile), 0) == 0
group_of = np.empty(nrun, dtype=np.int32); cap = int(nrun * nlive + nlive); out = np.empty(cap, dtype=np.int32); offs = np.zeros(nrun + 1, dtype=np.int64)
foc = [s for f, s in passes if f]
for mode, name in ((0, "fresh (host union-find)"), (1, "kept up to date")):
    L.mdns_core_set_incremental(h, 10 ** 9, 0)
    L.mdns_core_set_host_edges(h, 10 ** 12)
    best = None
    for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
        times = []
        for i, s in enumerate(foc):
            t0 = time.perf_counter()
            n = L.mdns_core_debug_groups(h, s.ctypes.data, len(s), mode, int(i == 0), group_of.ctypes.data, out.ctypes.data, cap, offs.ctypes.data)
            times.append(time.perf_counter() - t0)
            assert n > 0
        best = times if best is None or sum(times) < sum(best) else best
    print("%-24s first %.0f us, the other %d passes: mean %.0f us, max %.0f us, total %.1f ms"
          % (name, best[0] * 1e6, len(best) - 1, np.mean(best[1:]) * 1e6, np.max(best[1:]) * 1e6, sum(best) * 1e3))
# same results?
L.mdns_core_set_incremental(h, 10 ** 9, 1)
for i, s in enumerate(foc):
    assert L.mdns_core_debug_groups(h, s.ctypes.data, len(s), 1, int(i == 0), group_of.ctypes.data, out.ctypes.data, cap, offs.ctypes.data) > 0, L.mdns_core_last_error()
print("every pass: the same groups as a fresh computation")
