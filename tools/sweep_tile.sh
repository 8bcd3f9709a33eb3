# tile kernel (S sweeps per launch in LDS, tile edge T) against the streaming kernel (2 sweeps per pass) and the colour passes
for n in 128 256 512 1024 2048 4096; do
  SUHMO_GSRB_TILE=0 SUHMO_GSRB_VARIANT=0 python tools/gsrb_micro.py $n 64 5 | sed 's/^/colour passes      /'
  [ $n -ge 1024 ] && SUHMO_GSRB_TILE=0 SUHMO_GSRB_VARIANT=2 python tools/gsrb_micro.py $n 64 5 | sed 's/^/streaming K=2      /'
  for T in 16 32; do for S in 1 2 4; do
    SUHMO_GSRB_VARIANT=0 SUHMO_TILE_T=$T SUHMO_TILE_S=$S python tools/gsrb_micro.py $n 64 5 | sed "s/^/tile T=$T S=$S       /"
  done; done
done
