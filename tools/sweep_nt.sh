for n in 4096 2048; do for nt in 256 64; do for v in 1 2; do
  SUHMO_FUSED_MIN_CELLS=1 SUHMO_FUSED_NT=$nt SUHMO_GSRB_VARIANT=$v python tools/gsrb_micro.py $n 8 5 | sed "s/^/nt=$nt /"
done; done; done
