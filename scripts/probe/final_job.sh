set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --workload plummer1m --steps 20 --warmup 3 > gpurun_out/r02_final_plummer1m.json 2> gpurun_out/r02_final_plummer1m.err
timeout -k 10 600 python bench.py --workload box256k --steps 20 --warmup 3 > gpurun_out/r02_final_box256k.json 2> gpurun_out/r02_final_box256k.err
timeout -k 10 400 python scripts/bench_levels.py > gpurun_out/r02_final_levels.json 2> gpurun_out/r02_final_levels.err
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r02h
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02h -o plummer1m -- python3 $GRAFT_REPO_ROOT/bench.py --workload plummer1m --steps 20 --warmup 3 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_r02h_p.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02h -o box256k -- python3 $GRAFT_REPO_ROOT/bench.py --workload box256k --steps 20 --warmup 3 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_r02h_b.log 2>&1
for w in plummer1m box256k; do
  for g in fetch write sq; do
    case $g in
      fetch) pmc="FETCH_SIZE";;
      write) pmc="WRITE_SIZE";;
      sq) pmc="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU";;
    esac
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_r02c/$w/$g
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r02c/$w/$g -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 6 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/pmc_r02c_${w}_${g}.log 2>&1
    echo "pmc $w $g done"
  done
done
ls $GRAFT_REPO_ROOT/gpurun_out/prof_r02h | head
