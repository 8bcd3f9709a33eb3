# HBM bytes of ONE whole V-cycle of bench.py's workload, every kernel counted: rocprofv3 --pmc passes (one counter group per pass, no trace
# domains) over two runs that differ only in the number of timed V-cycles; what the longer run moved more, divided by the extra cycles, is a
# V-cycle's traffic (set-up, warm-up and the residual evaluations before / after cancel).
# usage (GPU box, repo root): bash tools/pmc_vcycle.sh <tag>        -> gpurun_out/<tag>_pmc_traffic_vcycle.json (+ per-kernel table .txt)
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_pmcv
KA=${KA:-4}; KB=${KB:-12}
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
for K in $KA $KB; do
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --output-format csv -d $OUT/k${K}_p$i -- python3 $R/bench.py --steps $K --warmup 2 --no-cpu --no-side > $OUT/k${K}_p$i.log 2>&1 || exit 1
    echo "pass K=$K $ctrs done"
  done
done
python3 $R/tools/make_traffic_json.py --vcycle $OUT $KA $KB $R/gpurun_out/${TAG}_pmc_traffic_vcycle.json > $R/gpurun_out/${TAG}_pmc_traffic_vcycle_by_kernel.txt
