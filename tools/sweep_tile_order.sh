# workgroup -> tile map of the tile kernel (SUHMO_TILE_ORDER: 0 as launched, 1 a contiguous run per XCD, 2 the same in panels of 8 tile rows):
# ms per sweep back to back at the depths' sizes, then the V-cycle of the bench
for n in 512 1024 2048; do
  for o in 0 1 2; do
    SUHMO_TILE_ORDER=$o python tools/gsrb_micro.py $n 64 5 | sed "s/^/tile order $o  /"
  done
done
for o in 0 1 2 0 2; do
  SUHMO_TILE_ORDER=$o python bench.py --steps 40 --warmup 10 --no-cpu --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tile order $o  bench: %.1f V-cycles/s  %.3f ms' % (d['value'], d['ms_per_step']))"
done
