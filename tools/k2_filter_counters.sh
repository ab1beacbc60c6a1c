#!/bin/bash
# SQ / cache counters of the K2 matrix-core filter (tools/muse_leg_probe.py launches it 22 times):
#   tools/k2_filter_counters.sh <out-dir under gpurun_out>
set -euo pipefail
repo=$(cd "$(dirname "$0")/.." && pwd)
out=gpurun_out/${1:?usage: tools/k2_filter_counters.sh <tag>}
export TMPDIR=/tmp
cd "$repo"
rm -rf "$out" && mkdir -p "$out"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d "$out/c$i" -- python3 tools/muse_leg_probe.py > "$out/c$i.log" 2>&1 || echo "counter set $i failed: $set"
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/c*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0].replace("void ", "").strip(), r["Counter_Name"])].append(float(r["Counter_Value"]))
res = collections.defaultdict(dict)
for (k, c), v in acc.items():
    res[k][c] = sum(v) / len(v)
json.dump(res, open(out + "/counters.json", "w"), indent=1, sort_keys=True)
for k in res:
    if "gemm" in k or "muse_rows" in k:
        print(k, json.dumps(res[k], sort_keys=True))
PY
find "$out" -name "*.csv" -delete
find "$out" -name "*.db" -delete
