# multi-rank sink tests (one gpurun call)
cd $GRAFT_REPO_ROOT
export GH_DD_DEBUG=${GH_DD_DEBUG_ON:+1}
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -k "sinks_on_ranks" > gpurun_out/sinks_mr.log 2>&1
grep -v "^\[dd\] rank . phase" gpurun_out/sinks_mr.log | grep "^\[dd\]\|^\[build\]" | awk '{ if ($0 ~ /held (4096|2048) cell count (4096|2048) spec 1: ->0: 0 <-0: 0 ->1: [0-9] /) next; print }' | tail -60
grep -v "^\[dd\]\|^\[build\]" gpurun_out/sinks_mr.log | tail -30
