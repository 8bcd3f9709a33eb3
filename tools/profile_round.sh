# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02
# 1. bench.py as the driver runs it; 2. rocprofv3 --kernel-trace --stats of the same command (no side configs, no CPU leg);
# 3. PMC passes (separate, no trace domains) for the HBM traffic of the relaxation kernels.  Everything lands under gpurun_out/<tag>_*.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R && python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 $R/bench.py --no-cpu --no-side > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_prof.err || exit 1
cp $(ls $OUT/${TAG}_prof/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
cd $R && python3 tools/stats_by_grid.py $(ls $OUT/${TAG}_prof/*/*kernel_trace.csv | head -1) 23 > $OUT/${TAG}_bench_kernel_stats_by_grid.txt 2>&1
bash $R/tools/pmc_any.sh ${TAG}_pmc k_gsrb_ bench.py --steps 6 --warmup 2 --no-cpu --no-side > $OUT/${TAG}_pmc_gsrb_kernels_summary.txt 2>&1
python3 $R/tools/make_traffic_json.py $OUT/${TAG}_pmc_gsrb_kernels_summary.txt 16777216 2 $OUT/${TAG}_pmc_traffic_gsrb.json > /dev/null
echo done
