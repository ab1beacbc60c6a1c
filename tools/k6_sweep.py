#!/usr/bin/env python3
"""K6 (bootstrapped radius) kernel time over pool sizes; MDNS_K6_SL=1|4|8|16 forces a shape.
python tools/k6_sweep.py"""
import ctypes as C, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import _lib
from massivedatans_amd.clustering import neighbors as nb
lib = _lib.require_device()
rng = np.random.RandomState(2)
out = {}
for K in (100, 400, 1000, 2000, 5000, 9000, 20000, 50000):
    pts = rng.uniform(size=(K, 3))
    np.random.seed(K)
    masks = nb.draw_bootstrap_masks(K, 10)
    s = nb.MemberSet(pts)
    r0 = s.bootstrap_radius_packed(masks, 10)
    lib.mdns_profile_every(1); lib.mdns_profile(8)
    for _ in range(20):
        r = s.bootstrap_radius_packed(masks, 10)
    n, ms = C.c_longlong(0), C.c_double(0)
    lib.mdns_profile_read(3, C.byref(n), C.byref(ms))
    lib.mdns_profile(0)
    assert r == r0
    out[K] = round(1e3 * ms.value / max(1, n.value), 1)
    s.close()
print(json.dumps({"MDNS_K6_SL": os.environ.get("MDNS_K6_SL", "auto"), "us_per_launch": out}))
