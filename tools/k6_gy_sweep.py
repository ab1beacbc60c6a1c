#!/usr/bin/env python3
"""K6 (k_nearest_uniform + k_nearest_finish) over the number of member chunks (grid.y), forced with
MDNS_K6_GY in a child process per value:   python tools/k6_gy_sweep.py
Prints one line per (K, gy): HIP-event time of the pair and the radius (which must not change)."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from massivedatans_amd.clustering import neighbors as nb
    from massivedatans_amd import _lib
    lib = _lib.require_device()
    for K in (1000, 2000, 5000, 9000, 20000, 50000):
        rng = np.random.RandomState(2)
        pts = rng.uniform(size=(K, 3))
        np.random.seed(K)
        masks = nb.draw_bootstrap_masks(K, 10)
        s = nb.MemberSet(pts)
        for _ in range(3):
            r = s.bootstrap_radius_packed(masks, 10)
        lib.mdns_profile_every(1); lib.mdns_profile(8)
        n_rep = 20 if K <= 9000 else 6
        for _ in range(n_rep):
            r = s.bootstrap_radius_packed(masks, 10)
        n, ms = C.c_longlong(0), C.c_double(0)
        lib.mdns_profile_read(3, C.byref(n), C.byref(ms)); lib.mdns_profile(0)
        print("gy", os.environ.get("MDNS_K6_GY", "auto"), "K", K, "us", round(ms.value * 1e3 / max(n.value, 1), 1), repr(r), flush=True)
else:
    for gy in ["auto", 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64]:
        env = dict(os.environ)
        if gy != "auto":
            env["MDNS_K6_GY"] = str(gy)
        subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
