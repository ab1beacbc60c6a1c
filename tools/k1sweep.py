import os, sys, json, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
from massivedatans_amd import _lib, gen
import bench_kernels as bk
lib = _lib.require_device()
d = gen.horns(10000); nx = 200
sp = lib.mdns_spectra_create(_lib.ptr(d["x"]), _lib.ptr(d["y"]), None, 10000, nx, 0)
rng = np.random.RandomState(1)
B = int(sys.argv[1])
cube = rng.uniform(size=(B, 3))
params = np.column_stack([10 ** (cube[:, 0] * 2 - 2), cube[:, 1] * 400 + 400, 10 ** (cube[:, 2] * 2)])
d_p = bk.dev(lib, params); d_L = lib.mdns_dev_alloc(B * 10000 * 8)
us = bk.timed(lib, 0, lambda: lib.mdns_gauss_loglike_batch_dev(sp, d_p, B, 0.01, None, 10000, d_L), reps=50)
print("B %d BT %s path %s: %.1f us  %.3e ev/s  issue_frac %.3f" % (B, os.environ.get("MDNS_K1_BT","auto"), os.environ.get("MDNS_K1_PATH","auto"), us, B*10000/(us*1e-6), 2*nx*B*10000*4/64/(us*1e-6)/(1024*2.4e9)))
