# A/B timing of library builds on the GPU box: bash scripts/probe/ab.sh <workload> <variant> ...   (variants/lib<variant>.so)
# The variant replaces the in-tree library of the box's scratch copy, because the host shell resolves libgandalf_hip.so by rpath.
w=$1; shift
for v in "$@"; do
  cp variants/lib$v.so gandalf_amd/csrc/libgandalf_hip.so
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err || { tail -3 gpurun_out/ab_$v.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("$v", round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["phase_ms_per_step"].items()})
PY
done
