#!/usr/bin/env python3
"""Integer bookkeeping of OUR host orchestration on the CPU oracle backends at BASELINE.json's
full C2 size (10 000 spectra, 100 live points), first 400 iterations: draw count, number of
accepted points and a SHA-256 of the pile of accepted points (their unit-cube coordinates come
from the RNG and the bit-exact geometry, so the pile is the same bytes exactly when every accept
decision and every RNG draw was the same).  The orchestration + oracle pair is
pinned bit for bit against the reference on the small traces (tests/test_orchestration.py); this
fixture extends the comparison of the GPU path with it to the full size, where the reference
itself takes hours.  Test infrastructure: writes tests/golden/bookkeeping_c2.json.

    python oracle/make_bookkeeping_hash.py [--all] [--case kind:ndata:nlive:iterations ...]

--case adds (or redoes) single entries, e.g. horns:100000:100:300 -- BASELINE.json configs[3]'s
data set, 100 000 spectra, on one host (a quarter of an hour on 8 cores).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from massivedatans_amd import gen, sample  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
import oracle_backend  # noqa: E402


class _Patch(object):
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def main():
    o = Oracle(kind="port-omp")
    oracle_backend.patch_neighbors(_Patch(), o)
    out = {}
    # the 700-iteration case (265 425 draws) takes 40 minutes on 8 cores; the others half a minute each
    cases = [("horns", 10000, 100, 400), ("nothing", 10000, 100, 400), ("horns", 10000, 100, 700)]
    path = os.path.join(ROOT, "tests", "golden", "bookkeeping_c2.json")
    if os.path.exists(path) and "--all" not in sys.argv:
        out = json.load(open(path))                  # keep what is there, redo the quick cases
        cases = cases[:2]
    if "--case" in sys.argv:
        cases = []
        for a in sys.argv[sys.argv.index("--case") + 1:]:
            kind, ndata, nlive, cap = a.split(":")
            cases.append((kind, int(ndata), int(nlive), int(cap)))
    for kind, ndata, nlive, cap in cases:
        # kind "horns-graph": the grouping of the reference's USE_GRAPH=1 default
        # (generate_subsets_graph: connected components, ids ascending), host implementation
        use_graph = kind.endswith("-graph")
        data = (gen.horns if kind.startswith("horns") else gen.nothing)(ndata)
        backend = oracle_backend.OracleSpectra(o, data["x"], data["y"])
        with np.errstate(all="ignore"):
            results, sampler, _, _ = sample.run(data["x"], data["y"], nlive_points=nlive, max_samples=cap,
                                                use_graph=use_graph, backend=backend)
        out["%s_%d_%d_%d" % (kind, ndata, nlive, cap)] = {
            "ndraws": int(sampler.ndraws), "npoints": int(len(sampler.pointpile)),
            "pointpile_sha256": hashlib.sha256(np.ascontiguousarray(sampler.pointpile, dtype=np.float64).tobytes()).hexdigest(),
            "logZ_first5": [float(v) for v in results["logZ"][:5]]}
        print(kind, out["%s_%d_%d_%d" % (kind, ndata, nlive, cap)])
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
