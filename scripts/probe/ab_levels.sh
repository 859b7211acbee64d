# A/B timing of library builds with the block-timestep bench: bash scripts/probe/ab_levels.sh <variant> ...   (variants/lib<variant>.so)
for v in "$@"; do
  cp variants/lib$v.so gandalf_amd/csrc/libgandalf_hip.so
  timeout -k 10 400 python scripts/bench_levels.py > gpurun_out/abl_$v.json 2>gpurun_out/abl_$v.err || { tail -3 gpurun_out/abl_$v.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$v.json"))
print("$v", round(d["ms_per_base_step"],3), {k:round(x,3) for k,x in d["phase_ms_per_base_step"].items()})
PY
done
