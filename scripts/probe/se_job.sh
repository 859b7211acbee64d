cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle.py tests/test_gpu_boundary.py -x -q -k "sink or potmin or physical_units or hybrid or quickselect" > gpurun_out/se_tests.log 2>&1
tail -12 gpurun_out/se_tests.log
timeout -k 10 800 python scripts/bench_sinks.py --params bb_units_1600 --N 2000000 --steps 32 > gpurun_out/c5_2m.json 2> gpurun_out/c5_2m.err
cat gpurun_out/c5_2m.json; tail -2 gpurun_out/c5_2m.err
timeout -k 10 600 python scripts/bench_sinks.py --N 2000000 --steps 16 > gpurun_out/se_sinks2m.json 2>/dev/null; cat gpurun_out/se_sinks2m.json
