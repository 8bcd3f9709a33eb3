T='tests/test_gpu_multiproc.py::test_rank_strips_as_processes[hier_dist.py-args7-4-10000]'
for f in test_gpu_hier_strips.py test_gpu_cfg5.py test_gpu_amr.py test_gpu_checkpoint.py test_gpu_hier_timestep.py test_gpu_moulin.py; do
  s=$(date +%s)
  python -m pytest tests/$f "$T" -m gpu -q -p no:cacheprovider --durations=2 > gpurun_out/bis_$f.txt 2>&1
  e=$(date +%s)
  echo "$f + args7: $((e-s)) s: $(grep -E 'args7|passed|failed' gpurun_out/bis_$f.txt | tr '\n' ' ')"
done
