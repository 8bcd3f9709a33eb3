# usage (GPU box, repo root): bash tools/probes/tile_probe.sh <tag>
# What bounds k_gsrb_tile<4, 32>?  A probe build of the library (in /tmp, with tools/probes/suhmo_gsrb_tile_probe.hip -- a copy of suhmo_gsrb.hip that carries the
# SUHMO_TILE_PROBE branches -- in the place of the product file, which has none) times the
# 4-sweep launch at 2048^2 with parts of the kernel switched off (env SUHMO_TILE_DBG: 1 no global loads, 2 no passes, 4 no barrier
# between passes, 8 (almost) no stores, 32 passes without LDS reads, 64 without LDS writes).  Results of the probe runs are wrong on purpose.
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/probe && cp -r $R/suhmo_amd $R/include $R/tools /tmp/probe/ && cd /tmp/probe/suhmo_amd/csrc || exit 1
cp /tmp/probe/tools/probes/suhmo_gsrb_tile_probe.hip suhmo_gsrb.hip || exit 1     # (the copy may lag behind the product kernel: it is a probe, not a mirror)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DSUHMO_TILE_PROBE -shared suhmo_level.hip suhmo_ops.hip suhmo_bcoef.hip suhmo_gsrb.hip suhmo_fas.hip suhmo_step.hip suhmo_rccl.hip suhmo_amr.hip suhmo_hier.hip suhmo_b2.hip suhmo_agg.hip suhmo_ipc.hip -o libsuhmo_hip.so -ldl || exit 1
cd /tmp/probe
for dbg in 0 2 1 9 11 13 41 105; do
  echo "SUHMO_TILE_DBG=$dbg"
  for n in 2048 1024; do SUHMO_TILE_DBG=$dbg SUHMO_GSRB_VARIANT=0 SUHMO_TILE_MAX_CELLS=100000000 python3 tools/gsrb_micro.py $n 8 8; done
done > $R/gpurun_out/${TAG}_tile_probe.txt 2>&1
cat $R/gpurun_out/${TAG}_tile_probe.txt
