cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -q -x -k "other_modes and bb_units" > gpurun_out/modes_tests.log 2>&1
tail -25 gpurun_out/modes_tests.log
