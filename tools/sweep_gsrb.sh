for cfg in "0 0" "1 0" "1 32" "1 64" "1 128" "1 256" "2 0" "2 64" "2 128" "2 256"; do
  set -- $cfg
  echo "variant=$1 hc=$2"
  SUHMO_GSRB_VARIANT=$1 SUHMO_FUSED_HC=$2 python bench.py --steps 5 --warmup 1 --sweeps-only 40 --no-cpu | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  vcycles/s %.1f  ms/step %.3f  sweep_ms(evt) %.4f  frac %.3f  bare: %.4f ms/sweep' % (d['value'], d['ms_per_step'], d['roofline']['avg_sweep_ms'], d['roofline']['frac'], d['bare_gsrb']['wall_ms_per_sweep']))"
done
