# depth-1 (2048^2) and depth-2 (1024^2) relaxation: kernel variant / threads per workgroup / rows per chunk
for n in 2048 1024; do
  for v in 0; do SUHMO_GSRB_VARIANT=$v python tools/gsrb_micro.py $n 8 5; done
  for nt in 256 64; do for v in 1 2; do for hc in 0 8 16 24 32 48 64; do
    SUHMO_FUSED_MIN_CELLS=1 SUHMO_FUSED_NT=$nt SUHMO_GSRB_VARIANT=$v SUHMO_FUSED_HC=$hc python tools/gsrb_micro.py $n 8 5 | sed "s/^/nt=$nt /"
  done; done; done
done
