#!/bin/bash
# Everything profiles/ keeps for a round, collected on the GPU box into gpurun_out/<tag>/:
#   tools/collect_round.sh r03
# (bench lines, complete analyses in both grouping modes + cProfile, rocprofv3 kernel stats of the
# bench and of a real run, PMC passes, K6 sweep and counters, chunk round trips)
set -euxo pipefail
repo=$(cd "$(dirname "$0")/.." && pwd)
cd "$repo"
tag=${1:?usage: tools/collect_round.sh <tag>}
out=gpurun_out/$tag
mkdir -p "$out"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
python bench.py --workload nothing --no-e2e-full > "$out/bench_nothing.json" 2> "$out/bench_nothing.err"
python bench.py --workload muse > "$out/bench_muse_b1.json" 2> "$out/bench_muse_b1.err"
python bench.py --workload muse --batch 64 --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > "$out/bench_muse_b64.json" 2> "$out/bench_muse_b64.err"
USE_GRAPH=1 python tools/e2e_run.py horns 10000 100 0 > "$out/e2e_c2_full_graph.json" 2> "$out/e2e_c2_full_graph.err"
USE_GRAPH=0 python tools/e2e_run.py horns 10000 100 0 > "$out/e2e_c2_full.json" 2> "$out/e2e_c2_full.err"
USE_GRAPH=1 python tools/e2e_run.py nothing 10000 100 0 > "$out/e2e_c3_full_graph.json" 2> "$out/e2e_c3_full_graph.err"
USE_GRAPH=0 python tools/e2e_run.py nothing 10000 100 0 > "$out/e2e_c3_full.json" 2> "$out/e2e_c3_full.err"
USE_GRAPH=1 MDNS_E2E_PROFILE=1 python tools/e2e_run.py horns 10000 100 0 > "$out/e2e_c2_full_graph_profiled.json" 2> "$out/e2e_c2_full_graph_profile.txt"
USE_GRAPH=0 MDNS_E2E_PROFILE=1 python tools/e2e_run.py horns 10000 100 0 > "$out/e2e_c2_full_profiled.json" 2> "$out/e2e_c2_full_profile.txt"
python tools/k6_sweep.py > "$out/k6_sweep.json" 2> "$out/k6_sweep.err"
MDNS_K6_PATH=classic python tools/k6_sweep.py > "$out/k6_sweep_classic.json" 2> "$out/k6_sweep_classic.err"
python tools/chunk_bench.py > "$out/chunk_bench.log" 2>&1
python tools/chunk_bench.py classic > "$out/chunk_bench_classic.log" 2>&1
export TMPDIR=/tmp
USE_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/e2e600_graph" -- python3 tools/e2e_run.py horns 10000 100 600 > "$out/e2e600_graph.log" 2>&1
USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/e2e600" -- python3 tools/e2e_run.py horns 10000 100 600 > "$out/e2e600.log" 2>&1
# (only 64 MiB travel back: the per-dispatch traces stay here, the --stats summaries go)
find "$out" -name "*_kernel_trace.csv" -delete
find "$out" -name "*.db" -delete
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d "$out/k6_5000/sq$i" -- python3 tools/k6_one.py 5000 > "$out/k6_5000_sq$i.log" 2>&1 || echo "counter set $i failed"
done
find "$out" -name "*_kernel_trace.csv" -delete
# the accept pass of issue-bound chunks: chain kernel against the matrix-core filter
for shape in "10000 256" "10000 128" "50000 256" "10000 1024"; do
    for m in 0 mfma; do MDNS_K1_FILTER=$m python tools/filter_bench.py $shape 2>&1 | tail -1; done
done > "$out/filter_bench.jsonl"
for p in 1 2 3; do MDNS_FILTER_PROBE=$p MDNS_K1_FILTER=mfma python tools/filter_bench.py 10000 256 2>&1 | tail -1; done > "$out/filter_probes.jsonl"
python tools/k6_gy_sweep.py > "$out/k6_gy_sweep.log" 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_probe.hip -o /tmp/mfma_probe 2>/dev/null && /tmp/mfma_probe > "$out/mfma_f64_probe.log" 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o /tmp/valu_probe 2>/dev/null && /tmp/valu_probe > "$out/valu_rate_probe.log" 2>&1
# round 4: the MUSE analysis end to end, the K2 matrix-core filter (round trips by shape, SQ / cache counters, the f64
# matrix rate by occupancy), the two-rank rehearsal with the evidence integration sharded
python tools/e2e_muse.py 6250 4096 40 600 > "$out/e2e_muse600.json" 2> "$out/e2e_muse600.err"
for shape in "6250 4096 64" "6250 4096 32" "2000 4096 57" "700 4096 40" "6250 1000 64" "777 333 17"; do
    python tools/k2_filter_bench.py $shape 20 2>&1 | tail -1
done > "$out/k2_filter_bench.jsonl"
tools/k2_filter_counters.sh "$tag/k2_counters" > "$out/k2_filter_counters.log" 2>&1 || echo "k2 counters failed"
hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_occupancy.hip -o /tmp/mfma_occ 2>/dev/null && /tmp/mfma_occ > "$out/mfma_f64_occupancy.log" 2>&1
tools/spmd_rehearsal.sh 25000 100 300 1 > "$out/spmd_rehearsal.log" 2>&1 || echo "rehearsal failed"
rm -rf gpurun_out/spmd
USE_GRAPH=1 python tools/e2e_run.py horns 100000 100 450 > "$out/e2e_c4_450_graph.json" 2> "$out/e2e_c4_450_graph.err"
find "$out" -name "*.db" -delete
du -sh "$out"
echo collected "$out"
