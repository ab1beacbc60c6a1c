"""One K1 shape, timed with HIP events: python tools/k1sweep.py <B> [ndata]
(env MDNS_K1_PATH=rows|cols and MDNS_K1_BT=1..16 force the kernel / candidate tile)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import _lib, gen
import bench_kernels as bk

lib = _lib.require_device()
B = int(sys.argv[1])
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
nx = 200
if nd == 10000:
    d = gen.horns(nd)
    sp = lib.mdns_spectra_create(_lib.ptr(d["x"]), _lib.ptr(d["y"]), None, nd, nx, 0)
else:
    y = np.ascontiguousarray(np.random.RandomState(nd).normal(0, 0.01, size=(nd, nx)))
    sp = lib.mdns_spectra_create(_lib.ptr(gen.wavelength_grid()), _lib.ptr(y), None, nd, nx, 1)
rng = np.random.RandomState(1)
cube = rng.uniform(size=(B, 3))
params = np.column_stack([10 ** (cube[:, 0] * 2 - 2), cube[:, 1] * 400 + 400, 10 ** (cube[:, 2] * 2)])
d_p = bk.dev(lib, params)
d_L = lib.mdns_dev_alloc(B * nd * 8)
us = bk.timed(lib, 0, lambda: lib.mdns_gauss_loglike_batch_dev(sp, d_p, B, 0.01, None, nd, d_L), reps=50 if nd <= 100000 else 10)
print("B %d nd %d BT %s path %s: %.1f us  %.3e ev/s  issue_frac %.3f  phys %.0f GB/s" % (
    B, nd, os.environ.get("MDNS_K1_BT", "auto"), os.environ.get("MDNS_K1_PATH", "auto"), us, B * nd / (us * 1e-6),
    2 * nx * B * nd * 4 / 64 / (us * 1e-6) / (1024 * 2.4e9), (nd * nx * 8 + B * nd * 8) / (us * 1e-6) / 1e9))
