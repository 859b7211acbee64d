cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_oracle.py -x -q -k "potmin" > gpurun_out/pm_tests.log 2>&1
tail -8 gpurun_out/pm_tests.log
