#!/bin/bash
# multi-rank rehearsal profile: scripts/probe/mr_job.sh TAG [env assignments...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
env THR_SERIAL=1 "$@" true
for kv in "$@"; do export "$kv"; done
THR_SERIAL=1 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o p -- python3 scripts/probe/threaded_ranks.py 8 1048576 4 plummer_4k > gpurun_out/${tag}_thr8.log 2>&1
grep "world 8" gpurun_out/${tag}_thr8.log | head -3
python3 scripts/mr_profile_summary.py gpurun_out/prof_$tag/p_results.db 8 4 > gpurun_out/${tag}_summary.txt
rm -rf gpurun_out/prof_$tag
grep "per rank per step\|k_let_walk" gpurun_out/${tag}_summary.txt
