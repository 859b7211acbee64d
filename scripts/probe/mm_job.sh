cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oracle.py -x -q -k "sinks or sink_run or potmin" > gpurun_out/mm_tests.log 2>&1
tail -3 gpurun_out/mm_tests.log
timeout -k 10 400 python scripts/bench_sinks.py > gpurun_out/mm_sinks.json 2> gpurun_out/mm_sinks.err
timeout -k 10 600 python scripts/bench_sinks.py --N 2000000 --steps 16 > gpurun_out/mm_sinks2m.json 2> gpurun_out/mm_sinks2m.err
python3 -c "
import json
for f in ('mm_sinks','mm_sinks2m'):
    j=json.load(open('gpurun_out/%s.json'%f)); print(f, round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['phase_ms_per_step'].items()}, j['N_end'], j['sink_Ngas'])"
