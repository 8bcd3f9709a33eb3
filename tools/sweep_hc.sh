for v in 1 2; do for hc in 16 24 32 37 40 44 48 49 50 52 56 60 64 72 80 96; do
  SUHMO_GSRB_VARIANT=$v SUHMO_FUSED_HC=$hc python tools/gsrb_micro.py 4096 8 5
done; done
