# usage (on the GPU box, from the repo root): bash tools/profile_hier.sh r03 [base cells per side] [steps]
# rocprofv3 --kernel-trace --stats of the cfg5 time step (tools/hier_bench.py: base + 3 AMR levels of box unions): per-kernel and
# per-kernel-and-grid statistics land under gpurun_out/<tag>_hier<base>_*
TAG=${1:-r03}
NB=${2:-4096}
NS=${3:-5}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_hier${NB}_prof -- python3 $R/tools/hier_bench.py $NB $NS > $OUT/${TAG}_hier${NB}_under_rocprof.txt 2> $OUT/${TAG}_hier${NB}_prof.err || exit 1
cp $(ls $OUT/${TAG}_hier${NB}_prof/*/*kernel_stats.csv | head -1) $OUT/${TAG}_hier${NB}_kernel_stats.csv
cd $R && python3 tools/stats_by_grid.py $(ls $OUT/${TAG}_hier${NB}_prof/*/*kernel_trace.csv | head -1) $((NS + 3)) > $OUT/${TAG}_hier${NB}_kernel_stats_by_grid.txt 2>&1
python3 tools/trace_around.py $(ls $OUT/${TAG}_hier${NB}_prof/*/*kernel_trace.csv | head -1) fillBufferAligned > $OUT/${TAG}_hier${NB}_around_fill.txt 2>&1
rm -rf $OUT/${TAG}_hier${NB}_prof
echo done
