cd $GRAFT_REPO_ROOT
for w in 16384 6000 2049; do
  GH_QSEL_WIDE_MIN=$w timeout -k 10 400 python scripts/bench_sinks.py --N 2000000 --steps 12 > gpurun_out/qw_w$w.json 2> gpurun_out/qw_w$w.err
  python3 -c "
import json,sys
j=json.load(open('gpurun_out/qw_w$w.json')); print($w, round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['phase_ms_per_step'].items()}, round(j['setup_s'],1))"
done
GH_QSEL_WIDE_MIN=6000 timeout -k 10 400 python scripts/bench_sinks.py > gpurun_out/qw_262k_w6000.json 2>/dev/null; python3 -c "
import json
j=json.load(open('gpurun_out/qw_262k_w6000.json')); print('262k w6000', round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['phase_ms_per_step'].items()})"
GH_QSEL_WIDE_MIN=2049 timeout -k 10 400 python scripts/bench_sinks.py > gpurun_out/qw_262k_w2049.json 2>/dev/null; python3 -c "
import json
j=json.load(open('gpurun_out/qw_262k_w2049.json')); print('262k w2049', round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['phase_ms_per_step'].items()})"
