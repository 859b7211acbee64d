cd $GRAFT_REPO_ROOT
timeout -k 10 400 python scripts/bench_sinks.py > gpurun_out/qw_sinks.json 2> gpurun_out/qw_sinks.err
cat gpurun_out/qw_sinks.json
timeout -k 10 600 python scripts/bench_sinks.py --N 2000000 --steps 16 > gpurun_out/qw_sinks2m.json 2> gpurun_out/qw_sinks2m.err
cat gpurun_out/qw_sinks2m.json
