#!/bin/bash
# Collects, on the GPU box, what profiles/ keeps per round for one bench workload:
#   kernel-trace --stats, FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md, HBM
#   section: they cannot share a pass), and a few SQ / scalar-cache counter passes.
# usage: tools/collect_profiles.sh <out-dir under gpurun_out> [bench.py arguments ...]
# (rocprofv3 runs python3 directly: no env / shell hop between the profiler and the program)
set -e
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out" && mkdir -p "$out"
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-hbm-leg $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py $ARGS > "$out/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py $ARGS > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py $ARGS > "$out/write.log" 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d "$out/sq$i" -- python3 bench.py $ARGS > "$out/sq$i.log" 2>&1 || echo "counter set $i failed: $set"
done
echo collected "$out"
