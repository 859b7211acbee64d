cd $GRAFT_REPO_ROOT
for n in 2 4; do
  GH_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus $n --steps 4 --warmup 1 --no-cpu > gpurun_out/benchmr_$n.json 2> gpurun_out/benchmr_$n.err || tail -5 gpurun_out/benchmr_$n.err
  python3 -c "
import json
j=json.load(open('gpurun_out/benchmr_$n.json'))
print($n, j['n_gpus'], round(j['ms_per_step'],2), j['config'].get('parallelism'), j.get('transport'), j.get('collectives_per_step'), (j.get('imbalance') or {}).get('max_over_mean'))"
done
