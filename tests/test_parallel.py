"""The N > 1 path on CPU: two processes, gloo backend, the oracle as per-rank scorer."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from massivedatans_amd import gen, parallel
    from oracle.oracle import Oracle
    from oracle_backend import OracleSpectra
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        orc = Oracle("port")
        d = gen.horns(37)                                   # ragged split: 19 + 18
        sharded = parallel.ShardedGaussLine(d["x"], d["y"], lambda x, y: OracleSpectra(orc, x, y))
        whole = OracleSpectra(orc, d["x"], d["y"])
        rng = np.random.RandomState(4)
        params = np.column_stack([rng.uniform(0.01, 1, 5), rng.uniform(400, 800, 5), 10 ** rng.uniform(0, 2, 5)])
        ok = True
        for mask in (np.ones(37, bool), rng.uniform(size=37) < 0.4, np.arange(37) < 3, np.arange(37) == 30):
            got = sharded.loglike_batch(params, mask)
            ok &= np.array_equal(got, whole.loglike_batch(params, mask))
        # live-point pool: ranks hold overlapping id sets of different sizes, twice (capacity reuse / growth)
        pile = rng.uniform(size=(500, 3))
        for k in (40, 300):
            ids = np.unique(rng.randint(0, 500, size=k + 7 * rank))
            uniq, pts = parallel.allgather_pool(ids, pile[ids])
            allids = [None] * world
            dist.all_gather_object(allids, ids)
            want = np.unique(np.concatenate(allids))
            ok &= np.array_equal(uniq, want) and np.array_equal(pts, pile[want])
        lo, hi = parallel.shard_range(37, rank, world)
        # the WHOLE analysis, SPMD: every rank runs the same host orchestration (same seed),
        # scores only its block of spectra and all-gathers the likelihood columns; the result
        # must be the reference trace, bit for bit, on every rank
        from massivedatans_amd import sample
        from massivedatans_amd.multi_nested_integrator import multi_nested_integrator
        import oracle_backend

        class _MP(object):
            def setattr(self, obj, name, val):
                setattr(obj, name, val)

        oracle_backend.patch_neighbors(_MP(), orc)
        from tracing import Recorder, check_bookkeeping, check_floats, load_trace
        g = load_trace("horns12")
        d12 = gen.horns(12)
        if os.environ.get("MDNS_LONG_TESTS") == "1":
            # (the classic form -- likelihood columns all-gathered per call, Python constrainer -- is the slow third
            # of this test: with MDNS_LONG_TESTS=1)
            backend = parallel.ShardedGaussLine(d12["x"], d12["y"], lambda x, y: OracleSpectra(orc, x, y))
            problem = sample.GaussLineProblem(d12["x"], d12["y"], backend=backend)
            sampler = sample.build_sampler(problem, nlive_points=int(g["nlive"]), nsuperset_draws=int(g["nsuperset_draws"]),
                                           use_graph=False, seed=1, batched=True)
            rec = Recorder(sampler)
            with np.errstate(all="ignore"):
                res = multi_nested_integrator(tolerance=0.5, multi_sampler=rec, min_samples=0,
                                              max_samples=int(g["max_samples"]))
            ok &= np.random.uniform() == float(g["rng_probe"])
            check_bookkeeping(g, sampler, rec, res)
            check_floats(g, rec, res, rtol=0)
        # the same with the joint state sharded (parallel.ShardedJointState over per-rank numpy
        # states): per draw chunk the ranks exchange B accept flags (MAX all-reduce) and the
        # accepted candidate's block -- not L[B, M] -- and still reproduce the reference trace
        backend = parallel.ShardedGaussLine(d12["x"], d12["y"], lambda x, y: OracleSpectra(orc, x, y))
        problem = sample.GaussLineProblem(d12["x"], d12["y"], backend=backend)
        sampler = sample.build_sampler(problem, nlive_points=int(g["nlive"]), nsuperset_draws=int(g["nsuperset_draws"]),
                                       use_graph=False, seed=1, batched=True, fused=True)
        ok &= type(sampler.joint).__name__ == "ShardedJointState"
        ok &= sampler.native is not None                  # one native call per draw on every rank, chunks meet in draw_params
        rec = Recorder(sampler)
        with np.errstate(all="ignore"):
            res = multi_nested_integrator(tolerance=0.5, multi_sampler=rec, min_samples=0,
                                          max_samples=int(g["max_samples"]))
        ok &= np.random.uniform() == float(g["rng_probe"])
        check_bookkeeping(g, sampler, rec, res)
        check_floats(g, rec, res, rtol=0)
        # the same again with the evidence integration sharded too (parallel.LocalColumns): every rank
        # integrates its own columns from its own joint state, the ranks agree on finished data sets at
        # every check, and the gathered evidences are the reference's, bit for bit
        np.random.seed(1)
        backend = parallel.ShardedGaussLine(d12["x"], d12["y"], lambda x, y: OracleSpectra(orc, x, y))
        problem = sample.GaussLineProblem(d12["x"], d12["y"], backend=backend)
        sampler = sample.build_sampler(problem, nlive_points=int(g["nlive"]), nsuperset_draws=int(g["nsuperset_draws"]),
                                       use_graph=False, seed=1, batched=True, fused=True)
        view = parallel.LocalColumns(sampler)
        ok &= view.ndata == parallel.shard_range(12, rank, world)[1] - parallel.shard_range(12, rank, world)[0]
        with np.errstate(all="ignore"):
            res = view.gather(multi_nested_integrator(tolerance=0.5, multi_sampler=view, min_samples=0,
                                                      max_samples=int(g["max_samples"])))
        ok &= np.random.uniform() == float(g["rng_probe"])
        for key in ("logZ", "logZerr", "information"):
            ok &= np.array_equal(res[key], g[key])
        ok &= res["columns"] == parallel.shard_range(12, rank, world) and len(res["weights"][0][2]) == view.joint.hi - view.joint.lo
        q.put((rank, bool(ok), lo, hi))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [p.exitcode for p in procs] == [0, 0]
    assert res == [(0, True, 0, 19), (1, True, 19, 37)]


def _muse_worker(rank, world, port, q, cases=("muse6", "muse10_graph"), local_columns=False):
    """configs[4]'s problem with the data sets sharded: the MUSE-style likelihood + its jitter as
    constrained draws over a ShardedJointState, every rank's native constrainer drawing the same
    noise stream and adding the columns of its own block."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from massivedatans_amd import gen, musefuse, parallel, sample
    from massivedatans_amd.multi_nested_integrator import multi_nested_integrator
    from oracle.oracle import Oracle
    import oracle_backend
    from oracle_backend import OracleMuseSpectra
    from tracing import Recorder, check_bookkeeping, check_floats, load_trace
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        orc = Oracle("port")

        class _MP(object):
            def setattr(self, obj, name, val):
                setattr(obj, name, val)

        oracle_backend.patch_neighbors(_MP(), orc)
        ok = True
        for case in cases:
            g = load_trace(case)
            data = gen.muse_like(int(g["ndata"]), int(g["nx"]))
            backend = parallel.ShardedMuse(data["x"], data["y"], data["v"], lambda x, y, v: OracleMuseSpectra(orc, x, y, v))
            problem = musefuse.MuseProblem(data["x"], data["y"], data["v"], backend=backend, jitter=True)
            sampler = sample.build_sampler(problem, nlive_points=int(g["nlive"]), nsuperset_draws=int(g["nsuperset_draws"]),
                                           use_graph=bool(g["use_graph"]), seed=1, batched=False, fused=True, native=True)
            ok &= type(sampler.joint).__name__ == "ShardedJointState" and sampler.native is not None
            if local_columns:
                # the evidence integration sharded too: every rank its own columns (here blocks of 2, 2, 1, 1 data sets:
                # ranks with ONE column of an analysis with several), the gathered evidences are the reference's
                view = parallel.LocalColumns(sampler)
                with np.errstate(all="ignore"):
                    res = view.gather(multi_nested_integrator(tolerance=0.5, multi_sampler=view, min_samples=0, max_samples=int(g["max_samples"])))
                sampler.native.sync_gauss_to_numpy()
                ok &= np.random.uniform() == float(g["rng_probe"])
                for key in ("logZ", "logZerr", "information"):
                    ok &= np.array_equal(res[key], g[key])
                ok &= int(sampler.ndraws) == int(g["ndraws"])
                continue
            rec = Recorder(sampler)
            with np.errstate(all="ignore"):
                res = multi_nested_integrator(tolerance=0.5, multi_sampler=rec, min_samples=0, max_samples=int(g["max_samples"]))
            sampler.native.sync_gauss_to_numpy()
            ok &= np.random.uniform() == float(g["rng_probe"])
            check_bookkeeping(g, sampler, rec, res)
            check_floats(g, rec, res, rtol=0)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_muse_gloo(world):
    """BASELINE configs[4] sharded: reference traces muse6 / muse10_graph bit for bit
    -- integers, floats, position of the random stream -- with 2 and 3 ranks (ragged blocks); with 4 ranks
    (blocks of 2, 2, 1, 1) and the evidence integration sharded as well (parallel.LocalColumns)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    # (the longer trace, 100 s through gloo, with MDNS_LONG_TESTS=1: ragged blocks of 4 + 3 + 3 data sets)
    cases = ("muse6", "muse10_graph") if os.environ.get("MDNS_LONG_TESTS") == "1" else ("muse6",)
    procs = [ctx.Process(target=_muse_worker, args=(r, world, port, q, cases, world == 4)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert [p.exitcode for p in procs] == [0] * world
    assert res == [(r, True) for r in range(world)]


def test_shard_bounds():
    from massivedatans_amd import parallel
    assert parallel.shard_bounds(10, 3).tolist() == [0, 4, 7, 10]
    assert parallel.shard_bounds(100000, 8).tolist() == [12500 * k for k in range(9)]
    assert parallel.shard_bounds(3, 8).tolist() == [0, 1, 2, 3, 3, 3, 3, 3, 3]
    m = np.arange(10) % 2 == 0
    assert parallel.local_mask(m, 1, 3).tolist() == m[4:7].tolist()


_GPU_RANK_SCRIPT = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch                      # before libmdns_hip: ONE HIP runtime per process (DESIGN.md 6)
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
from massivedatans_amd import gen, parallel, sample
from massivedatans_amd.like import GaussLineSpectra
from massivedatans_amd.multi_nested_integrator import multi_nested_integrator
d = gen.horns(300)
os.environ["MDNS_CORE_HOST_EDGES"] = "0"      # every grouping on the device
backend = parallel.ShardedGaussLine(d["x"], d["y"], lambda x, y: GaussLineSpectra(x, y, noise_level=0.01))
out = {}
for name, fused, use_graph in (("classic", False, False), ("fused", True, False), ("graph", True, True)):
    problem = sample.GaussLineProblem(d["x"], d["y"], backend=backend)
    sampler = sample.build_sampler(problem, nlive_points=40, use_graph=use_graph, seed=1, batched=True, fused=fused)
    with np.errstate(all="ignore"):
        res = multi_nested_integrator(tolerance=0.5, multi_sampler=sampler, min_samples=0, max_samples=120)
    out[name] = dict(ndraws=int(sampler.ndraws), npoints=int(len(sampler.pointpile)),
                     pile=np.ascontiguousarray(sampler.pointpile).tobytes().hex()[:64],
                     logZ=[float(v) for v in res["logZ"][:5]], joint=type(sampler.joint).__name__,
                     device_groups=sampler._dgroups is not None and sampler._dgroups.ncalls > 0)
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_with_hip_kernels(tmp_path):
    """The N-rank analysis with the REAL kernels: two fresh child processes (never an exec from a
    process that touched the GPU) share the one GPU, each holds half of the spectra
    (parallel.ShardedGaussLine over like.GaussLineSpectra) and, in the fused form, half of the
    joint state (parallel.ShardedJointState over jointstate.GaussJointState); they exchange through
    gloo (RCCL wants one device per rank).  Both ranks, both forms and the single-process run
    must take the same draws."""
    import json
    import subprocess
    script = tmp_path / "rank.py"
    script.write_text(_GPU_RANK_SCRIPT)
    port = str(29600 + os.getpid() % 300)
    env = dict(os.environ, MDNS_DEVICE="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][0][7:]) for so, _ in outs]
    assert res[0] == res[1]
    assert res[0]["fused"]["joint"] == "ShardedJointState" and res[0]["classic"]["joint"] == "NoneType"
    for k in ("ndraws", "npoints", "pile"):
        assert res[0]["fused"][k] == res[0]["classic"][k]
    assert np.allclose(res[0]["fused"]["logZ"], res[0]["classic"]["logZ"], rtol=0, atol=1e-9)
    # and the single process
    from massivedatans_amd import gen, sample
    d = gen.horns(300)
    with np.errstate(all="ignore"):
        results, sampler, _, _ = sample.run(d["x"], d["y"], nlive_points=40, max_samples=120, use_graph=False)
    assert sampler.ndraws == res[0]["fused"]["ndraws"] and len(sampler.pointpile) == res[0]["fused"]["npoints"]
    assert np.ascontiguousarray(sampler.pointpile).tobytes().hex()[:64] == res[0]["fused"]["pile"]
    assert np.allclose(results["logZ"][:5], res[0]["fused"]["logZ"], rtol=0, atol=1e-9)
    # the reference's default grouping, components computed on the device by every rank
    assert res[0]["graph"]["device_groups"] and res[0]["graph"]["joint"] == "ShardedJointState"
    with np.errstate(all="ignore"):
        results, sampler, _, _ = sample.run(d["x"], d["y"], nlive_points=40, max_samples=120, use_graph=True)
    assert sampler.ndraws == res[0]["graph"]["ndraws"] and len(sampler.pointpile) == res[0]["graph"]["npoints"]
    assert np.ascontiguousarray(sampler.pointpile).tobytes().hex()[:64] == res[0]["graph"]["pile"]
    assert np.allclose(results["logZ"][:5], res[0]["graph"]["logZ"], rtol=0, atol=1e-9)


_GPU_MUSE_SCRIPT = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch                      # before libmdns_hip: ONE HIP runtime per process
import torch.distributed as dist
world = int(sys.argv[4])
dist.init_process_group(sys.argv[5], init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=world,
                        **({"device_id": torch.device("cuda", 0)} if sys.argv[5] == "nccl" else {}))
if sys.argv[5] == "nccl":
    import ctypes
    from massivedatans_amd import _lib
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    _lib.check(_lib.require_device().mdns_set_stream(ctypes.c_void_p(stream.cuda_stream)), "mdns_set_stream")
from massivedatans_amd import gen, musefuse, parallel, sample
from massivedatans_amd.like import GaussLineSpectra, MuseSpectra
out = {}
d = gen.muse_like(10, 512)
backend = parallel.ShardedMuse(d["x"], d["y"], d["v"], lambda x, y, v: MuseSpectra(x, y, v))
with np.errstate(all="ignore"):
    res, sampler, _, _ = musefuse.run(d["x"], d["y"], d["v"], nlive_points=30, max_samples=80, use_graph=True, backend=backend)
out["muse"] = dict(ndraws=int(sampler.ndraws), npoints=int(len(sampler.pointpile)), joint=type(sampler.joint).__name__,
                   local=type(sampler.joint.local).__name__, direct=sampler.joint._direct() is not None,
                   pile=np.ascontiguousarray(sampler.pointpile).tobytes().hex()[:64], logZ=[float(v) for v in res["logZ"][:5]],
                   probe=float(np.random.uniform()))
g = gen.horns(300)
backend = parallel.ShardedGaussLine(g["x"], g["y"], lambda x, y: GaussLineSpectra(x, y, noise_level=0.01))
with np.errstate(all="ignore"):
    res, sampler, _, _ = sample.run(g["x"], g["y"], nlive_points=40, max_samples=120, use_graph=True, backend=backend)
out["gauss"] = dict(ndraws=int(sampler.ndraws), npoints=int(len(sampler.pointpile)), joint=type(sampler.joint).__name__,
                    direct=sampler.joint._direct() is not None,
                    pile=np.ascontiguousarray(sampler.pointpile).tobytes().hex()[:64], logZ=[float(v) for v in res["logZ"][:5]])
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _single_process_references():
    from massivedatans_amd import gen, musefuse, sample
    d = gen.muse_like(10, 512)
    with np.errstate(all="ignore"):
        res, sampler, _, _ = musefuse.run(d["x"], d["y"], d["v"], nlive_points=30, max_samples=80, use_graph=True)
    muse = dict(ndraws=int(sampler.ndraws), npoints=int(len(sampler.pointpile)),
                pile=np.ascontiguousarray(sampler.pointpile).tobytes().hex()[:64], logZ=res["logZ"][:5], probe=float(np.random.uniform()))
    g = gen.horns(300)
    with np.errstate(all="ignore"):
        res, sampler, _, _ = sample.run(g["x"], g["y"], nlive_points=40, max_samples=120, use_graph=True)
    gauss = dict(ndraws=int(sampler.ndraws), npoints=int(len(sampler.pointpile)),
                 pile=np.ascontiguousarray(sampler.pointpile).tobytes().hex()[:64], logZ=res["logZ"][:5])
    return muse, gauss


def _same_run(got, want):
    assert got["ndraws"] == want["ndraws"] and got["npoints"] == want["npoints"] and got["pile"] == want["pile"]
    assert np.allclose(got["logZ"], want["logZ"], rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_sharded_muse_two_ranks_share_the_gpu(tmp_path):
    """configs[4]'s problem sharded with the REAL kernels: two child processes share
    the one GPU (gloo), each holds half of the spectra + variances (parallel.ShardedMuse over
    like.MuseSpectra) and half of the joint state (ShardedJointState over MuseJointState: the chunk in two
    halves on the device, mdns_backend_draw_score / _commit, the votes MAX-reduced in between); the noise
    of musefuse.py:535 is drawn by every rank from the common stream.  Same draws, accepted points, evidences and
    position of the random stream as the single process; the Gaussian-line problem through the same halves."""
    import json
    import subprocess
    script = tmp_path / "muse_rank.py"
    script.write_text(_GPU_MUSE_SCRIPT)
    port = str(29950 + os.getpid() % 300)
    env = dict(os.environ, MDNS_DEVICE="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), "2", "gloo"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][0][7:]) for so, _ in outs]
    assert res[0] == res[1]
    assert res[0]["muse"]["joint"] == "ShardedJointState" and res[0]["muse"]["local"] == "MuseJointState"
    muse, gauss = _single_process_references()
    _same_run(res[0]["muse"], muse)
    assert res[0]["muse"]["probe"] == muse["probe"]
    _same_run(res[0]["gauss"], gauss)


@pytest.mark.gpu
def test_sharded_states_over_direct_rccl_with_one_rank(tmp_path):
    """The path an N-GPU run takes, as far as one GPU allows: backend nccl with ONE
    rank, the library's kernels on torch's stream, ShardedJointState reducing the votes with
    ncclAllReduce called directly on that stream (massivedatans_amd/rccl.py) between the two halves of
    every chunk.  Same results as the plain single process, for both problems."""
    import json
    import subprocess
    script = tmp_path / "muse_rank.py"
    script.write_text(_GPU_MUSE_SCRIPT)
    port = str(30250 + os.getpid() % 300)
    out = subprocess.run([sys.executable, str(script), ROOT, port, "0", "1", "nccl"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MDNS_DEVICE="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert res["muse"]["direct"] and res["gauss"]["direct"], "RCCL was not called directly"
    muse, gauss = _single_process_references()
    _same_run(res["muse"], muse)
    _same_run(res["gauss"], gauss)


def test_rccl_binding_matches_the_header():
    """massivedatans_amd/rccl.py against rccl.h: the id is 128 bytes passed by value, the enum
    values are the header's, and the entry points the sharded path calls exist."""
    import ctypes
    import re
    from massivedatans_amd import rccl
    assert ctypes.sizeof(rccl._UniqueId) == rccl.UNIQUE_ID_BYTES == 128
    L = rccl.lib()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclAllReduce", "ncclGetErrorString"):
        assert hasattr(L, name)
    header = "/opt/rocm/include/rccl/rccl.h"
    if os.path.exists(header):
        text = open(header).read()
        for name, value in (("ncclInt32", rccl.INT32), ("ncclInt64", rccl.INT64), ("ncclUint64", rccl.UINT64),
                            ("ncclFloat64", rccl.FLOAT64), ("ncclMax", rccl.MAX), ("ncclSum", rccl.SUM), ("ncclMin", rccl.MIN)):
            m = re.search(r"\b%s\s*=\s*(\d+)" % name, text)
            assert m and int(m.group(1)) == value, name
        assert re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+128", text)


_RCCL_SCRIPT = r'''
import sys, ctypes as C
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch                                  # first: one HIP runtime per process
torch.cuda.set_device(0)
from massivedatans_amd import _lib, rccl
lib = _lib.require_device()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
_lib.check(lib.mdns_set_stream(C.c_void_p(stream.cuda_stream)), "stream")
comm = rccl.Communicator(1, 0, lambda payload: payload)
n = 1000
a = np.arange(n, dtype=np.int32) % 7
bits = np.arange(n, dtype=np.int64) * 3
d_a, d_b, d_c = lib.mdns_dev_alloc(4 * n), lib.mdns_dev_alloc(8 * n), lib.mdns_dev_alloc(8 * n)
_lib.check(lib.mdns_h2d(d_a, _lib.ptr(a), a.nbytes), "h2d")
_lib.check(lib.mdns_h2d(d_b, _lib.ptr(bits), bits.nbytes), "h2d")
comm.all_reduce(d_a, d_a, n, rccl.INT32, rccl.MAX, stream.cuda_stream)
comm.all_gather(d_b, d_c, n, rccl.INT64, stream.cuda_stream)
a2, c = np.empty_like(a), np.empty_like(bits)
_lib.check(lib.mdns_d2h(_lib.ptr(a2), d_a, a.nbytes), "d2h")
_lib.check(lib.mdns_d2h(_lib.ptr(c), d_c, c.nbytes), "d2h")
assert np.array_equal(a2, a) and np.array_equal(c, bits)
comm.destroy()
print("RCCL OK")
'''


@pytest.mark.gpu
def test_direct_rccl_on_the_library_stream(tmp_path):
    """massivedatans_amd/rccl.py on hardware, as far as one GPU allows: a one-rank communicator,
    MAX all-reduce in place and an all-gather on library-owned device memory, on the stream the
    kernels use.  (bench.py --gpus N uses the same calls; N > 1 needs N devices.)"""
    import subprocess
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_SCRIPT)
    out = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, MDNS_DEVICE="0"))
    assert out.returncode == 0 and "RCCL OK" in out.stdout, out.stderr[-2000:]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher must start two ranks itself (torch.distributed.run
    on 127.0.0.1) -- on a box without GPUs that ends, in the RANKS, with "no HIP device", not with a
    request to be launched differently."""
    import subprocess
    if _gpu_visible():
        pytest.skip("a GPU is visible: the ranks would run the bench (covered by the -m gpu test below)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300)
    text = out.stdout + out.stderr
    assert out.returncode != 0
    assert "must be launched with" not in text
    assert "No HIP GPUs are available" in text or "no HIP device" in text, text[-1500:]


def _gpu_visible():
    from massivedatans_amd import _lib
    try:
        return _lib.load().mdns_device_count() > 0
    except _lib.MdnsError:
        return False


@pytest.mark.gpu
def test_bench_step_with_collectives_on_one_gpu():
    """bench.py's distributed step on hardware as far as one GPU allows: MDNS_BENCH_FORCE_DIST=1 makes a
    one-rank process group, so the step runs its RCCL exchanges (pool all-gather, MAX all-reduce of the
    accept flags in place, all-gather of the fill bits) on the kernels' stream, and the line keeps the
    contract."""
    import json
    import subprocess
    env = dict(os.environ, MDNS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-e2e",
                          "--no-hbm-leg", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["value"] > 0
    assert line["config"]["collectives"], line["config"]


@pytest.mark.gpu
def test_bench_muse_step_with_collectives_on_one_gpu():
    """`bench.py --workload muse --gpus N`'s step as far as one GPU allows: a chunk of a
    constrained draw over the sharded MUSE-style joint state -- templates + K2 + accept test, MAX
    all-reduce of the votes on the kernels' stream, commit, all-gather of the fill bits -- with one rank."""
    import json
    import subprocess
    env = dict(os.environ, MDNS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 200),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "muse", "--batch", "8", "--muse-ndata", "1000",
                          "--steps", "5", "--warmup", "2", "--no-e2e", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["value"] > 0
    assert "constrained draw" in line["config"]["step"] and line["config"]["collectives"] != "none", line["config"]
