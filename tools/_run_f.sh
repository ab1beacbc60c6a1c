timeout -k 10 600 python -m pytest tests/test_orchestration.py tests/test_muse.py -m gpu -x -q > gpurun_out/r3_gputest_r.log 2>&1; tail -3 gpurun_out/r3_gputest_r.log
USE_GRAPH=1 python tools/e2e_run.py horns 10000 100 0 > gpurun_out/e2e_r_1.json 2> gpurun_out/e2e_r_1.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/e2e_r_1.json"))
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ("wall_s","ndraws","draw_constrained_wall_s")})
nc=d["native_constrainer"]; print({k:round(v/1e9,2) for k,v in nc.items() if k.startswith("ns_")})
PY
