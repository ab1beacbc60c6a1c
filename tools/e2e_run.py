#!/usr/bin/env python3
"""End-to-end analysis on the GPU: python tools/e2e_run.py <horns|nothing> <ndata> <nlive> <max_samples>
(max_samples 0 = run to the termination criterion; MDNS_E2E_PROFILE=1 adds a cProfile summary on stderr;
USE_GRAPH=1 takes the reference's default grouping, connected components, computed on the device)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import gen, sample

kind, ndata, nlive, cap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
data = (gen.horns if kind == "horns" else gen.nothing)(ndata)
t0 = time.time()
# a heartbeat on stderr for long runs (the GPU queue treats a silent process as hung)
import threading
from massivedatans_amd import multi_nested_sampler as _mns
_live = {}
_orig_init = _mns.MultiNestedSampler.__init__
def _init(self, *a, **k):
    _orig_init(self, *a, **k)
    _live["sampler"] = self
_mns.MultiNestedSampler.__init__ = _init
def _beat():
    while not _live.get("done"):
        time.sleep(60)
        s = _live.get("sampler")
        if s is not None and not _live.get("done"):
            print("[%.0f s] iteration %d, %d data sets running, %d draws" % (time.time() - t0, s.global_iter, s.ndata, s.ndraws),
                  file=sys.stderr, flush=True)
threading.Thread(target=_beat, daemon=True).start()
def _go():
    with np.errstate(all="ignore"):
        return sample.run(data["x"], data["y"], nlive_points=nlive, max_samples=cap,
                          use_graph=os.environ.get("USE_GRAPH", "0") == "1",
                          fused=os.environ.get("MDNS_FUSED", "1") != "0")
if os.environ.get("MDNS_E2E_PROFILE") == "1":        # cProfile of the whole run, top entries on stderr
    import cProfile, pstats
    prof = cProfile.Profile()
    results, sampler, problem, duration = prof.runcall(_go)
    pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(30)
else:
    results, sampler, problem, duration = _go()
_live["done"] = True
if sampler._dgroups is not None and sampler._dgroups.size_log:
    log = np.array(sampler._dgroups.size_log)
    edges = [2, 8, 32, 128, 512, 2048, 8192, 1 << 30]
    hist, lo = {}, 0
    for e in edges:
        pick = (log[:, 0] >= lo) & (log[:, 0] < e)
        hist["M<%d" % e] = {"calls": int(pick.sum()), "split": int((log[pick, 1] > 1).sum())}
        lo = e
    print("grouping calls by selection size:", json.dumps(hist), file=sys.stderr)
print(json.dumps({"workload": "%s %d x 200, nlive %d, cap %d" % (kind, ndata, nlive, cap), "wall_s": duration,
                  "setup_s": time.time() - t0 - duration, "iterations": int(results["nsamples"]),
                  "ndraws": int(sampler.ndraws), "evals_useful": int(sampler.nevals),
                  "evals_scored": int(problem.nevals + (sampler.joint.nevals_scored if sampler.joint is not None else 0)
                                      + (sampler.native.stats()["pairs"] if sampler.native is not None else 0)),
                  "launches": int(problem.ncalls + (sampler.joint.ncalls if sampler.joint is not None else 0)),
                  "fused": sampler.joint is not None, "constrained_draws": int(sampler.ndraw_calls),
                  "draw_chunks": int(sampler.ndraw_chunks + (sampler.native.stats()["chunks"] if sampler.native is not None else 0)),
                  "native_constrainer": sampler.native.stats() if sampler.native is not None else None,
                  "draw_constrained_wall_s": sampler.draw_seconds,
                  "grouping": ("graph (components on the device, %d calls)" % sampler._dgroups.ncalls) if sampler._dgroups is not None
                  else ("graph (host)" if sampler.use_graph else "walk (host)"),
                  "core": sampler.core_stats() if hasattr(sampler, "core_stats") else None,
                  "fill_wall_s": getattr(sampler, "fill_seconds", None),
                  "useful_evals_per_s": sampler.nevals / duration,
                  "logZ_first3": results["logZ"][:3].tolist()}))
