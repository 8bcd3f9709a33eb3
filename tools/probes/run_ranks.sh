#!/bin/bash
# usage: run_ranks.sh <world> <seconds> <tool.py> [args...]   -- N gloo ranks of a tools/*.py on this box's GPU, each under a timeout; a rank
# that is still running after <seconds> - 20 dumps the Python stack of every thread (where it waits) before it is killed.  Logs: gpurun_out/rank_<r>.log
W=$1; T=$2; TOOL=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
PORT=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((W - 1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT SUHMO_DIST_BACKEND=gloo OMP_NUM_THREADS=1 \
  timeout -k 5 $T python3 -c "
import faulthandler, runpy, sys
faulthandler.dump_traceback_later(max(1, $T - 20), exit=False)
sys.argv = ['$TOOL'] + '$*'.split()
runpy.run_path('$R/tools/$TOOL', run_name='__main__')
" > $R/gpurun_out/rank_$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
echo "exit code $rc"
exit $rc
