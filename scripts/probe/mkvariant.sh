# Quick A/B library: bash scripts/probe/mkvariant.sh <name> <file.hip> [extra hipcc flags]
# Recompiles ONE translation unit with only the 3-D M4 instantiations (GH_PROBE_3D_M4) and links it with the in-tree
# objects of the others into variants/lib<name>.so (timed on the box by scripts/probe/ab.sh).  Not a product build.
set -e
name=$1; f=$2; shift 2
cd "$(dirname "$0")/../../gandalf_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DGH_PROBE_3D_M4 "$@" -c $f -o /tmp/probe_${f%.hip}.o 2>/tmp/probe_build.log
objs=""
for o in api tree density forces gravity integrate nbody stars cd2010 comm sinks rccl_comm; do
  if [ "$o.hip" = "$f" ]; then objs="$objs /tmp/probe_$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -ldl -o ../../variants/lib$name.so
grep -A9 "k_grav_evalILi3ELb0ELi4ELi0ELi0E" /tmp/probe_build.log | grep -E "VGPRs|Scratch|Occupancy|Spill" || true
