for v in A B C A B C; do
  GANDALF_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib$v.so timeout -k 10 300 python bench.py --workload plummer1m --steps 20 --warmup 3 --no-cpu > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("$v", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["phase_ms_per_step"].items()})
PY
done
