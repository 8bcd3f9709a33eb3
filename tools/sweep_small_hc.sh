# small depths: fused kernel with short row chunks vs the colour-pass kernel
for n in 1024 512 256; do
  SUHMO_GSRB_VARIANT=0 python tools/gsrb_micro.py $n 8 5
  for v in 1 2; do for hc in 4 6 8 12 16 24; do
    SUHMO_FUSED_MIN_CELLS=1 SUHMO_GSRB_VARIANT=$v SUHMO_FUSED_HC=$hc python tools/gsrb_micro.py $n 8 5
  done; done
done
