# copies the outputs of scripts/probe/final_job.sh (gpurun_out/) into profiles/r03_* and prints the numbers DESIGN section 5 quotes
cd "$(dirname "$0")/../.."
R=r03
cp gpurun_out/${R}_final_plummer1m.json profiles/${R}_bench_plummer1m.json
cp gpurun_out/${R}_final_box256k.json profiles/${R}_bench_box256k.json
cp gpurun_out/${R}_final_levels.json profiles/${R}_bench_levels_plummer1m.json
cp gpurun_out/${R}_final_sinks.json profiles/${R}_bench_sinks_262k.json
cp gpurun_out/${R}_final_sinks2m.json profiles/${R}_bench_sinks_2m.json
cp gpurun_out/prof_${R}/plummer1m_kernel_stats.csv profiles/${R}_plummer1m_kernel_stats.csv
cp gpurun_out/prof_${R}/box256k_kernel_stats.csv profiles/${R}_box256k_kernel_stats.csv
cp gpurun_out/${R}mr_summary.txt profiles/${R}_multirank_8x1m_kernels.txt
python scripts/pmc_to_json.py plummer1m ${R} gpurun_out/pmc_${R}/plummer1m > /dev/null
python scripts/pmc_to_json.py box256k ${R} gpurun_out/pmc_${R}/box256k > /dev/null
python - <<'PY'
import json
for w in ("plummer1m","box256k"):
    j=json.load(open("profiles/r03_bench_%s.json"%w))
    print(w, round(j["ms_per_step"],3), "%.4e"%j["value"], "cpu %.3e x%.0f"%(j["cpu_baseline"]["value"], j["value"]/j["cpu_baseline"]["value"]), {k:round(v,3) for k,v in j["phase_ms_per_step"].items()})
    print("  roof", round(j["roofline"]["frac"],3), "dens", round(j["roofline_density"]["frac"],3), round(j["roofline_density"]["avg_launch_ms"],3))
for f in ("levels_plummer1m","sinks_262k","sinks_2m"):
    j=json.load(open("profiles/r03_bench_%s.json"%f)); print(f, round(j.get("ms_per_base_step", j.get("ms_per_step")),3), {k:round(v,2) for k,v in j.get("phase_ms_per_base_step", j.get("phase_ms_per_step")).items()})
print(json.load(open("profiles/pmc_traffic.json"))["_library_sha16"])
PY
sha256sum gandalf_amd/csrc/libgandalf_hip.so | cut -c1-16
head -1 profiles/${R}_multirank_8x1m_kernels.txt
