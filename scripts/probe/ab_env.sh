# one variant, several values of an environment variable: bash scripts/probe/ab_env.sh <workload> <variant> <VAR> <v1> <v2> ...
w=$1; v=$2; var=$3; shift 3
cp variants/lib$v.so gandalf_amd/csrc/libgandalf_hip.so
for x in "$@"; do
  env $var=$x timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu > gpurun_out/abe_$x.json 2>gpurun_out/abe_$x.err || { tail -3 gpurun_out/abe_$x.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abe_$x.json"))
print("$var=$x", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["phase_ms_per_step"].items()})
PY
done
