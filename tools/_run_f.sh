timeout -k 10 1100 python -m pytest tests/test_orchestration.py tests/test_muse.py tests/test_hip_parity.py tests/test_abi.py -m gpu -x -q > gpurun_out/r3_gputest_o.log 2>&1; tail -4 gpurun_out/r3_gputest_o.log
for g in 1 0; do
USE_GRAPH=$g python tools/e2e_run.py horns 10000 100 0 > gpurun_out/e2e_o_$g.json 2> gpurun_out/e2e_o_$g.err; python - $g <<'PY'
import json,sys
d=json.load(open("gpurun_out/e2e_o_%s.json"%sys.argv[1]))
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ("wall_s","ndraws","draw_constrained_wall_s","grouping")})
nc=d["native_constrainer"]; print({k:round(v/1e9,2) for k,v in nc.items() if k.startswith("ns_")})
PY
done
