cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_oracle.py -x -q -k "physical_units or examples_boss or restart" > gpurun_out/un_tests.log 2>&1
tail -30 gpurun_out/un_tests.log
