#!/usr/bin/env python3
"""BASELINE.json configs[2] (gennothing) or configs[1] (horns), 10 000 spectra and 100 live points,
run TO TERMINATION by our host orchestration on the CPU oracle backends: evidences of all data
sets, draw count and a SHA-256 of the pile of accepted points.  gennothing: 2 minutes on 8
cores; horns: an hour or two (its regions hold thousands of points for several hundred iterations).  The pair orchestration +
oracle is pinned bit for bit against the reference on the small traces; this fixture lets the GPU
path be compared with it over a complete run at full size.  Test infrastructure: writes
tests/golden/full_c3.npz (nothing) or full_c2.npz (horns).

    python oracle/make_full_run.py [nothing|horns] [graph]

`graph`: the reference's default grouping (USE_GRAPH=1: connected components, host implementation)
instead of the discovery-order walk -> full_c3_graph.npz / full_c2_graph.npz.
"""
import hashlib
import os
import sys
import time

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
    kind = sys.argv[1] if len(sys.argv) > 1 else "nothing"
    graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
    data = (gen.nothing if kind == "nothing" else gen.horns)(10000)
    backend = oracle_backend.OracleSpectra(o, data["x"], data["y"])
    t = time.time()
    with np.errstate(all="ignore"):
        results, sampler, _, _ = sample.run(data["x"], data["y"], nlive_points=100, max_samples=0,
                                            use_graph=graph, backend=backend)
    digest = hashlib.sha256(np.ascontiguousarray(sampler.pointpile, dtype=np.float64).tobytes()).hexdigest()
    name = ("full_c3" if kind == "nothing" else "full_c2") + ("_graph" if graph else "") + ".npz"
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name),
                        logZ=results["logZ"], logZerr=results["logZerr"], ndraws=sampler.ndraws,
                        npoints=len(sampler.pointpile), iterations=results["nsamples"],
                        pointpile_sha256=np.array(digest))
    print("iterations %d, ndraws %d, npoints %d, %s, %.0f s" % (results["nsamples"], sampler.ndraws,
                                                                len(sampler.pointpile), digest[:16], time.time() - t))


if __name__ == "__main__":
    main()
