cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python scripts/probe/threaded_ranks.py 8 2000000 8 bb_units_1600 > gpurun_out/c5_thr8_2m.log 2>&1
tail -8 gpurun_out/c5_thr8_2m.log
