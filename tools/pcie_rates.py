#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rates of the boundary on configs[1]'s shape (10 000 x 200):
what a caller sees who hands over host arrays and takes host arrays back, next to the resident
rate bench.py reports.  One JSON line.   python tools/pcie_rates.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import _lib, gen, jointstate, sample
from massivedatans_amd.like import GaussLineSpectra

nd, B = 10000, 256
d = gen.horns(nd)
x, y = d["x"], np.ascontiguousarray(d["y"])
rng = np.random.RandomState(1)
params = np.column_stack([rng.uniform(0.01, 1, B), rng.uniform(400, 800, B), 10 ** rng.uniform(0, 2, B)])
mask = np.ones(nd, dtype=np.bool_)
out = {}


def rate(fn, evals, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    return {"evals_per_s": evals / dt, "us_per_call": dt * 1e6}


# 1. the zero-edit drop-in: the reference's own `like` signature (sample.py:85-96), host arrays in
#    and out, ONE candidate per call; the 16 MB of spectra travel with every call ...
shim = C.CDLL(os.path.join(_lib.DROPIN_DIR, "clike.so"))
shim.like.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
Lout = np.zeros(nd)
p = params[0]


def dropin():
    Lout[:] = 0
    shim.like(x.ctypes.data, y.ctypes.data, nd, len(x), p[0], p[1], p[2], 0.01, mask.ctypes.data, Lout.ctypes.data)


out["dropin_like_reupload"] = rate(dropin, nd, 20)
# ... unless the caller registers them once
lib = _lib.require_device()
_lib.check(lib.mdns_register_spectra(y.ctypes.data, None, nd, len(x)), "register")
out["dropin_like_registered"] = rate(dropin, nd, 200)
lib.mdns_unregister_spectra(y.ctypes.data)

# 2. the batched binding with host pointers: 256 candidates in, L[256, 10 000] (20 MB) back
sp = GaussLineSpectra(x, y)
out["batch_256_host_in_out"] = rate(lambda: sp.loglike_batch(params), B * nd, 20)
out["batch_1_host_in_out"] = rate(lambda: sp.loglike_batch(params[:1]), nd, 200)

# 3. the fused draw with host pointers: 256 candidates in, {index, fill bits} back
js = jointstate.GaussJointState(sp, 100, sample.kernel_params, fetch_rows=False)
js.init(sample.priortransform_batch(rng.uniform(size=(100, 3))))
js.prepare()
hopeless = np.column_stack([np.full(B, 10.0), rng.uniform(400, 800, size=B), np.full(B, 2.0)])   # (A, mu, log sig): nobody accepts
out["fused_draw_256_host_in_out"] = rate(lambda: js.draw(hopeless, None), B * nd, 200)
js.close()
sp.close()
print(json.dumps(out))
