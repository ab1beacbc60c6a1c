#!/usr/bin/env python3
"""The MUSE-style analysis (BASELINE.json configs[4]) end to end on the GPU:
python tools/e2e_muse.py <ndata> <nx> <nlive> <max_samples> [nojitter]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import gen, musefuse

ndata, nx, nlive, cap = (int(v) for v in sys.argv[1:5])
jitter = not (len(sys.argv) > 5 and sys.argv[5] == "nojitter")
t0 = time.time()
data = gen.muse_like(ndata, nx)
t1 = time.time()
with np.errstate(all="ignore"):
    results, sampler, problem, duration = musefuse.run(data["x"], data["y"], data["v"], nlive_points=nlive, max_samples=cap,
                                                      use_graph=os.environ.get("USE_GRAPH", "1") == "1", jitter=jitter)
st = sampler.native.stats() if sampler.native is not None else {}
print(json.dumps({"workload": "muse %d x %d, nlive %d, cap %d, jitter %s" % (ndata, nx, nlive, cap, jitter), "gen_s": t1 - t0,
                  "wall_s": duration, "iterations": int(results["nsamples"]), "ndraws": int(sampler.ndraws),
                  "constrained_draws": int(sampler.ndraw_calls), "evals_useful": int(sampler.nevals),
                  "draw_constrained_wall_s": sampler.draw_seconds,
                  "evals_per_s_in_draw_constrained": (int(sampler.nevals) - nlive * ndata) / sampler.draw_seconds if sampler.draw_seconds else None,
                  "native_constrainer": st, "logZ_first3": results["logZ"][:3].tolist()}))
