cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_oracle.py -x -q -k "restart or snapshot" > gpurun_out/rs_tests.log 2>&1
tail -25 gpurun_out/rs_tests.log
