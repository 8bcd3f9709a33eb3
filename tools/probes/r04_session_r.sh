cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SUHMO_GRAPH_MAX_CELLS=0 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_r_prof -o h -- python3 $R/tools/hier_bench.py 4096 4 > $R/gpurun_out/r04_r_prof.log 2>&1
cd $R
T=$(ls gpurun_out/r04_r_prof/*kernel_trace.csv gpurun_out/r04_r_prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_busy.py $T 0.4 > gpurun_out/r04_r_busy.txt 2>&1
python3 tools/stats_by_grid.py $T 308 > gpurun_out/r04_r_by_grid.txt 2>&1
head -3 gpurun_out/r04_r_busy.txt; tail -3 gpurun_out/r04_r_prof.log
rm -rf gpurun_out/r04_r_prof
