timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_n.log 2>&1; tail -4 gpurun_out/r3_gputest_n.log
USE_GRAPH=1 python tools/e2e_run.py horns 10000 100 0 > gpurun_out/e2e_n_graph.json 2> gpurun_out/e2e_n_graph.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/e2e_n_graph.json"))
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))})
nc=d["native_constrainer"]; print({k:round(v/1e9,2) for k,v in nc.items() if k.startswith("ns_")})
PY
