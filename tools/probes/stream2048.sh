# streaming K = 2 kernel at 2048^2 against the tile kernel (4 sweeps per launch): chunk heights around the one-round limit (2030 waves)
for hc in 19 20 21 22 23 24 26; do
  echo "NT=64 HC=$hc"
  SUHMO_GSRB_TILE=0 SUHMO_GSRB_VARIANT=2 SUHMO_FUSED_MIN_CELLS=1 SUHMO_FUSED_NT=64 SUHMO_FUSED_HC=$hc python3 tools/gsrb_micro.py 2048 8 8
done
