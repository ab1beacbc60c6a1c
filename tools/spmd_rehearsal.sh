#!/bin/bash
# Rehearsal of the N-rank whole-analysis path on ONE GPU: 2 ranks share the device and exchange
# through gloo (RCCL wants a device per rank); compared with the single-process run.  Prints, per run and
# rank, where the host time went: inside the sampler (draws + grouping: every rank repeats it) and in the
# evidence integration (sharded: every rank its own columns; MDNS_SHARD_INTEGRATION=0: all of them).
# usage: tools/spmd_rehearsal.sh [ndata] [nlive] [maxsamples] [use_graph]
set -euo pipefail
ND=${1:-1000}; NL=${2:-50}; MS=${3:-200}; UG=${4:-0}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/gpurun_out/spmd" && cd "$repo/gpurun_out/spmd"
export PYTHONPATH="$repo" NLIVE_POINTS=$NL MAXSAMPLES=$MS USE_GRAPH=$UG
python -c "from massivedatans_amd import gen; gen.save('d.npz', gen.horns($ND))"
base=d.npz_MLFRIENDS_nlive${NL}_${ND}.out8
rm -f d.npz_MLFRIENDS*; python -m massivedatans_amd.sample d.npz $ND
mv $base.npz single.npz; mv $base.stats.json single.stats.json
for shard in 1 0; do
    rm -f d.npz_MLFRIENDS*
    MDNS_SHARD_INTEGRATION=$shard MDNS_DIST_BACKEND=gloo MDNS_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29533 -m massivedatans_amd.sample d.npz $ND
    mkdir -p shard$shard && mv d.npz_MLFRIENDS* shard$shard/
done
python - <<PY
import glob, json, numpy as np
a = np.load('single.npz'); sa = json.load(open('single.stats.json'))
print(json.dumps({"run": "single process", "ndata": $ND, "wall_s": sa["duration"], "seconds": sa["seconds"], "fill_s": sa["fill_seconds"]}))
b = np.load('shard1/$base.evidence.npz')
parts = sorted(glob.glob('shard1/$base.cols*.stats.json'))
for f in parts:
    s = json.load(open(f))
    print(json.dumps({"run": "2 ranks, integration sharded", "columns": f.split('.cols')[1].split('.')[0], "wall_s": s["duration"],
                      "seconds": s["seconds"], "fill_s": s["fill_seconds"]}))
    assert int(s["ndraws"]) == int(sa["ndraws"])
c = np.load('shard0/$base.npz'); sc = json.load(open('shard0/$base.stats.json'))
print(json.dumps({"run": "2 ranks, every rank integrates everything", "wall_s": sc["duration"], "seconds": sc["seconds"], "fill_s": sc["fill_seconds"]}))
print('ndraws', int(a['ndraws']), int(c['ndraws']), 'max |dlogZ|', float(np.max(np.abs(a['logZ'] - b['logZ']))), float(np.max(np.abs(a['logZ'] - c['logZ']))))
assert int(a['ndraws']) == int(c['ndraws']) and np.max(np.abs(a['logZ'] - b['logZ'])) < 1e-9 and np.max(np.abs(a['logZ'] - c['logZ'])) < 1e-9
u = np.concatenate([np.load(f.replace('.stats.json', '.npz'))['L'] for f in parts], axis=1)
assert np.array_equal(u, a['L']), "posterior samples of the ranks' columns side by side are not the single run's"
print('SPMD rehearsal ok')
PY
