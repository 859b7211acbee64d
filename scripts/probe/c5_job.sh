cd $GRAFT_REPO_ROOT
timeout -k 10 800 python scripts/bench_sinks.py --params bb_units_1600 --N 2000000 --steps 32 > gpurun_out/c5_2m.json 2> gpurun_out/c5_2m.err
cat gpurun_out/c5_2m.json; tail -3 gpurun_out/c5_2m.err
