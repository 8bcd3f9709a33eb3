# usage (GPU box, repo root): bash tools/probes/bcoef_order_ab.sh  -- k_bcoef_fused with its tiles as launched (SUHMO_TILE_ORDER=0) and in XCD-contiguous runs (default)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; export TMPDIR=/tmp
cd /tmp
for o in 0 2; do
  SUHMO_TILE_ORDER=$o rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bco_$o -- python3 $R/bench.py --no-cpu --no-side > $OUT/bco_$o.json 2> $OUT/bco_$o.err
  echo "SUHMO_TILE_ORDER=$o: $(grep -h k_bcoef_fused $OUT/bco_$o/*/*kernel_stats.csv | head -1)"
  python3 -c "
import json
j=json.loads([l for l in open('$OUT/bco_$o.json') if l.startswith('{')][0]); print('   V-cycle under the profiler: %.3f ms' % j['ms_per_step'])"
  rm -rf $OUT/bco_$o
done
