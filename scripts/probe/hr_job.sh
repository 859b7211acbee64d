cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -k "lists_grow or sinks or quadrupole or gadget2 or eigenmac" > gpurun_out/hr_tests.log 2>&1
tail -15 gpurun_out/hr_tests.log
