cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_oracle.py -x -q -k "snapshot" > gpurun_out/sf_tests.log 2>&1
tail -5 gpurun_out/sf_tests.log
