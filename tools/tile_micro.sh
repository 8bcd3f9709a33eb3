# usage (GPU box, repo root): bash tools/tile_micro.sh <tag>   -- the tile kernel alone: ms per sweep at 2048^2, 1024^2, 512^2 (4 sweeps per launch)
TAG=${1:-x}
for n in 2048 1024 512; do
  SUHMO_GSRB_VARIANT=0 SUHMO_TILE_MAX_CELLS=100000000 python3 tools/gsrb_micro.py $n 8 8
done > gpurun_out/${TAG}_tile_micro.txt 2>&1
cat gpurun_out/${TAG}_tile_micro.txt
