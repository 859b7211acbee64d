cd $GRAFT_REPO_ROOT
timeout -k 10 600 python scripts/probe/threaded_ranks.py 8 65536 10 bb_sinks_8k_levels > gpurun_out/sinks_thr8.log 2>&1
tail -8 gpurun_out/sinks_thr8.log
GH_DD_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py -x -q -k "sinks_on_ranks and 8k-12-2" > gpurun_out/sinks_mr_dbg.log 2>&1
grep "speculative" gpurun_out/sinks_mr_dbg.log | sort | uniq -c | head; tail -3 gpurun_out/sinks_mr_dbg.log
