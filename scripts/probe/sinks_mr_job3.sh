cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_multirank.py -x -q > gpurun_out/mr_all.log 2>&1
tail -5 gpurun_out/mr_all.log
