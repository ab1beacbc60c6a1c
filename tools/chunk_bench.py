#!/usr/bin/env python3
"""Wall-clock of ONE draw chunk through the entry points a native constrainer calls
(mdns_backend_draw_begin / mdns_backend_draw_chunk), host call to polled outcome, for the shapes a
real run is made of:   python tools/chunk_bench.py [classic]
(candidates that no data set accepts, so the state never changes; MDNS_CHUNK_PATH=classic in the
environment -- or the argument -- takes the five-command path for comparison)"""
import json, os, sys, time
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "classic":
    os.environ["MDNS_CHUNK_PATH"] = "classic"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import gen, jointstate, sample
from massivedatans_amd.like import GaussLineSpectra

ndata, nlive = 10000, 100
d = gen.horns(ndata)
spectra = GaussLineSpectra(d["x"], d["y"], noise_level=0.01)
js = jointstate.GaussJointState(spectra, nlive, sample.kernel_params, fetch_rows=False, via_backend=True)
rng = np.random.RandomState(1)
cube = rng.uniform(size=(nlive, 3)); cube[:, 0] *= 0.01
js.init(sample.priortransform_batch(cube))
js.prepare()
out = []
for M in (1, 64, 300, 900, 3000, 4096, 10000):
    rows = None if M == ndata else np.sort(rng.choice(ndata, size=M, replace=False)).astype(np.int32)
    for B in (4, 32, 128):
        bad = np.column_stack([np.full(B, 1.0), rng.uniform(size=B), np.full(B, 1.0)])      # bright, broad: rejected everywhere
        params = sample.kernel_params(sample.priortransform_batch(bad))
        for _ in range(20):
            idx = js.draw_params(params, rows)[0]
        assert idx == -1
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            js.draw_params(params, rows)
        us = (time.perf_counter() - t0) / n * 1e6
        out.append({"M": M, "B": B, "us_per_chunk": round(us, 1)})
        print(out[-1], flush=True)
print(json.dumps({"path": os.environ.get("MDNS_CHUNK_PATH", "two launches"), "chunks": out}))
