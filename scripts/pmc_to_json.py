#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> profiles/pmc_traffic.json (what bench.py reports as `traffic` / measured instruction counts).

    python scripts/pmc_to_json.py <workload> <tag> <dir with the passes' output> [...]

Reads every *counter_collection.csv below the given directories (one directory per pass, or one for all), takes the
MEDIAN over the dispatches of each kernel (the first dispatches of a run belong to the setup, whose density passes start
from a guessed h), writes profiles/<tag>_<workload>_pmc.txt (all counters, per kernel) and updates
profiles/pmc_traffic.json[<workload>] with
    HBM bytes per launch   = 2 x FETCH_SIZE + WRITE_SIZE   (KiB -> bytes; the factor 2 is the gfx950 correction of
                             /opt/skills/guides/MI355X_MICROARCH.md, "HBM": FETCH_SIZE counts 128-byte requests at 64 B)
    fp64 lane-instructions = 64 x (SQ_INSTS_VALU_ADD_F64 + _MUL_F64 + _FMA_F64 + _TRANS_F64)
and the hash of the library the numbers were measured on (bench.py ignores them for any other build)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha16(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def median(v):
    v = sorted(v)
    return v[len(v)//2]


def main():
    workload, tag, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = ["# rocprofv3 --pmc passes of `python3 bench.py --workload %s` (separate passes per counter group); median over dispatches" % workload,
             "# units as rocprofv3 reports them: FETCH_SIZE / WRITE_SIZE in KiB, SQ_* in their own units (SQ_WAVE_CYCLES etc. count quad-cycles)"]
    med = {}
    for name in sorted(acc):
        if not any(k in name for k in ("k_dens", "k_density", "k_grav", "k_hydro")):
            continue
        lines.append(name)
        med[name] = {}
        for c in sorted(acc[name]):
            v = acc[name][c]
            med[name][c] = median(v)
            lines.append("   %-28s n=%d median=%.6g min=%.6g max=%.6g" % (c, len(v), median(v), min(v), max(v)))
    out_txt = os.path.join(ROOT, "profiles", "%s_%s_pmc.txt" % (tag, workload))
    open(out_txt, "w").write("\n".join(lines) + "\n")

    def hbm(prefix):
        tot = 0.0
        for name, c in med.items():
            if name.startswith(prefix) and "true" not in name.split("<")[1].split(",")[1:2]:
                tot += 2.0*c.get("FETCH_SIZE", 0.0)*1024.0 + c.get("WRITE_SIZE", 0.0)*1024.0
        return tot

    def first(prefix):
        for name in sorted(med):
            if name.startswith(prefix) and ", true" not in name:
                return med[name]
        return {}

    entry = {}
    ge = first("k_grav_eval<")
    if ge:
        entry["k_grav_eval"] = 2.0*ge.get("FETCH_SIZE", 0.0)*1024 + ge.get("WRITE_SIZE", 0.0)*1024
        f64 = sum(ge.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        if f64:
            entry["k_grav_eval_f64_lane_instr"] = 64.0*f64
            entry["k_grav_eval_f64_flops"] = 64.0*(f64 + ge.get("SQ_INSTS_VALU_FMA_F64", 0.0))     # an FMA counts 2, everything else 1
    gw = first("k_grav_walk<")
    if gw:
        entry["k_grav_walk"] = 2.0*gw.get("FETCH_SIZE", 0.0)*1024 + gw.get("WRITE_SIZE", 0.0)*1024
    hf = first("k_hydro_forces<")
    if hf:
        entry["k_hydro_forces"] = 2.0*hf.get("FETCH_SIZE", 0.0)*1024 + hf.get("WRITE_SIZE", 0.0)*1024
    dens = 0.0
    for pre in ("k_dens_walk<", "k_dens_eval<", "k_density<"):
        c = first(pre)
        dens += 2.0*c.get("FETCH_SIZE", 0.0)*1024 + c.get("WRITE_SIZE", 0.0)*1024
    entry["density_pass_hbm_bytes"] = dens
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        allp = json.load(open(pj))
    except (OSError, ValueError):
        allp = {}
    so = os.path.join(ROOT, "gandalf_amd", "csrc", "libgandalf_hip.so")
    if allp.get("_library_sha16") != sha16(so):
        allp = {}                      # numbers of another build: drop them all
    allp["_comment"] = ("per-launch medians from rocprofv3 --pmc passes (scripts/pmc_to_json.py; sources profiles/*_pmc.txt): HBM bytes = "
                        "2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the MI355X guide), fp64 lane-instructions = 64 x SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64")
    allp["_library_sha16"] = sha16(so)
    allp[workload] = entry
    json.dump(allp, open(pj, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
