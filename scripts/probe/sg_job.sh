cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_quickselect or sinks_match" > gpurun_out/sg_tests.log 2>&1
tail -5 gpurun_out/sg_tests.log
timeout -k 10 600 python scripts/bench_sinks.py --N 2000000 --steps 16 > gpurun_out/sg_sinks2m.json 2>/dev/null; cat gpurun_out/sg_sinks2m.json
timeout -k 10 400 python scripts/bench_sinks.py > gpurun_out/sg_sinks.json 2>/dev/null; cat gpurun_out/sg_sinks.json
