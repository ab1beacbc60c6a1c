#!/bin/bash
# Rehearsal of the N-rank whole-analysis path on ONE GPU: 2 ranks share the device and exchange
# through gloo (RCCL wants a device per rank); compared with the single-process run.
# usage: tools/spmd_rehearsal.sh [ndata] [nlive] [maxsamples]
set -e
ND=${1:-1000}; NL=${2:-50}; MS=${3:-200}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/spmd && cd gpurun_out/spmd
export PYTHONPATH=../.. NLIVE_POINTS=$NL MAXSAMPLES=$MS USE_GRAPH=0
python -c "from massivedatans_amd import gen; gen.save('d.npz', gen.horns($ND))"
rm -f d.npz_MLFRIENDS*; python -m massivedatans_amd.sample d.npz $ND; mv d.npz_MLFRIENDS_nlive${NL}_${ND}.out8.npz single.npz
MDNS_DIST_BACKEND=gloo MDNS_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 -m massivedatans_amd.sample d.npz $ND
python - <<PY
import numpy as np
a = np.load('single.npz'); b = np.load('d.npz_MLFRIENDS_nlive${NL}_${ND}.out8.npz')
print('ndraws', int(a['ndraws']), int(b['ndraws']), 'max |dlogZ|', float(np.max(np.abs(a['logZ'] - b['logZ']))))
assert int(a['ndraws']) == int(b['ndraws']) and np.max(np.abs(a['logZ'] - b['logZ'])) < 1e-9
print('SPMD rehearsal ok')
PY
