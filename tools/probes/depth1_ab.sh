# V-cycle at 4096^2 with depth 1 (2048^2) on the tile kernel (default until round 3) against the streaming kernel at its one-round chunk height
for tm in 8000000 3000000; do
  echo "SUHMO_TILE_MAX_CELLS=$tm"
  SUHMO_TILE_MAX_CELLS=$tm python3 bench.py --no-side --no-cpu 2>/dev/null | python3 -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])"
done
