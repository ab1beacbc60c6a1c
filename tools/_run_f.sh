timeout -k 10 900 python -m pytest tests/test_joint.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r3_gputest_q.log 2>&1; tail -4 gpurun_out/r3_gputest_q.log
python tools/k6_sweep.py 2>/dev/null | cut -c1-300
