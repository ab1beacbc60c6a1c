#!/usr/bin/env python3
"""The accept pass of an issue-bound chunk (10 000 spectra x 256 candidates by default), chain kernel
against the guarded filters:   MDNS_K1_FILTER=0|1|mfma python tools/filter_bench.py [ndata] [B]
Prints the wall time per chunk (score + commit, host call to polled outcome) and the HIP-event time
of the dominant accept kernel (class 0 of mdns_profile)."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import _lib, gen, jointstate, sample
from massivedatans_amd.like import GaussLineSpectra

ndata = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nlive = 100
d = gen.horns(ndata)
spectra = GaussLineSpectra(d["x"], d["y"], noise_level=0.01)
js = jointstate.GaussJointState(spectra, nlive, sample.kernel_params, fetch_rows=False)
rng = np.random.RandomState(1)
cube = rng.uniform(size=(nlive, 3))
js.init(sample.priortransform_batch(cube))
js.prepare()
lib = _lib.require_device()
out = {"filter": os.environ.get("MDNS_K1_FILTER", "default"), "flush": os.environ.get("FILTER_BENCH_FLUSH", "0"), "ndata": ndata, "B": B}
for name, bright in (("nobody accepts", True), ("random candidates", False)):
    cube = rng.uniform(size=(B, 3))
    if bright:
        cube[:, 0] = 1.0; cube[:, 2] = 1.0
    else:
        # candidates as a sampler draws them (narrow lines: template values down to denormals), against
        # thresholds nobody beats
        high = np.full((nlive, ndata), 1e300)                   # (kept alive across the call)
        _lib.check(lib.mdns_joint_set_live(js._h, _lib.ptr(high)), "set_live")
        js.prepare()
    xs = sample.priortransform_batch(cube)
    for _ in range(10):
        idx = js.draw(xs, None)[0]
        if idx >= 0:
            break
    if idx >= 0 and bright:
        raise SystemExit("a candidate was accepted")
    if idx >= 0:
        out[name] = "accepted %d: state changed, not timed" % idx
        continue
    lib.mdns_profile_every(1)
    lib.mdns_profile(1)
    n = 200
    flush = os.environ.get("FILTER_BENCH_FLUSH") == "1"
    if flush:
        # a sampler step runs other kernels between two accept passes: here a 128 MB device copy,
        # which leaves nothing of the spectra in the L2s
        nbytes = 128 << 20
        fa, fb = lib.mdns_dev_alloc(C.c_size_t(nbytes)), lib.mdns_dev_alloc(C.c_size_t(nbytes))
    t0 = time.perf_counter()
    for _ in range(n):
        if flush:
            lib.mdns_d2d(C.c_void_p(fa), C.c_void_p(fb), C.c_size_t(nbytes))
        js.draw(xs, None)
    wall = (time.perf_counter() - t0) / n * 1e6
    k, ms = C.c_longlong(0), C.c_double(0)
    lib.mdns_profile_read(0, C.byref(k), C.byref(ms))
    lib.mdns_profile(0)
    out[name] = {"wall_us_per_chunk": round(wall, 1), "kernel": (lib.mdns_profile_kernel(0) or b"").decode(),
                 "kernel_us": round(ms.value * 1e3 / max(k.value, 1), 2), "launches": k.value}
print(json.dumps(out))
