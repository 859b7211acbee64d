cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_quickselect or sinks_match" > gpurun_out/qw_tests.log 2>&1
tail -3 gpurun_out/qw_tests.log
bash scripts/probe/qw_job3.sh
