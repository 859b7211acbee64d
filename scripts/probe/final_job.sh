# round-3 measurement job (one gpurun call): benches, kernel stats, PMC passes, multi-rank rehearsal profile, level / sink benches
cd $GRAFT_REPO_ROOT
R=r03
timeout -k 10 600 python bench.py --workload plummer1m --steps 20 --warmup 5 > gpurun_out/${R}_final_plummer1m.json 2> gpurun_out/${R}_final_plummer1m.err
timeout -k 10 600 python bench.py --workload box256k --steps 20 --warmup 5 > gpurun_out/${R}_final_box256k.json 2> gpurun_out/${R}_final_box256k.err
timeout -k 10 400 python scripts/bench_levels.py > gpurun_out/${R}_final_levels.json 2> gpurun_out/${R}_final_levels.err
timeout -k 10 400 python scripts/bench_sinks.py > gpurun_out/${R}_final_sinks.json 2> gpurun_out/${R}_final_sinks.err
timeout -k 10 600 python scripts/bench_sinks.py --N 2000000 --steps 16 > gpurun_out/${R}_final_sinks2m.json 2> gpurun_out/${R}_final_sinks2m.err
echo "benches done"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_${R}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${R} -o plummer1m -- python3 $GRAFT_REPO_ROOT/bench.py --workload plummer1m --steps 20 --warmup 3 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_${R}_p.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${R} -o box256k -- python3 $GRAFT_REPO_ROOT/bench.py --workload box256k --steps 20 --warmup 3 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_${R}_b.log 2>&1
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_${R}/*kernel_trace.csv
echo "kernel stats done"
for w in plummer1m box256k; do
  for g in fetch write sq; do
    case $g in
      fetch) pmc="FETCH_SIZE";;
      write) pmc="WRITE_SIZE";;
      sq) pmc="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU";;
    esac
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_${R}/$w/$g
    rocprofv3 --pmc $pmc --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${R}/$w/$g -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 6 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/pmc_${R}_${w}_${g}.log 2>&1
    echo "pmc $w $g done"
  done
done
cd $GRAFT_REPO_ROOT
bash scripts/probe/mr_job.sh ${R}mr
