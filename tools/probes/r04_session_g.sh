cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_hier.py tests/test_gpu_hier_timestep.py tests/test_gpu_cfg5.py -m gpu -x -q -k "not north_star" > gpurun_out/r04_g_hier.log 2>&1
tail -3 gpurun_out/r04_g_hier.log
(python3 tools/hier_bench.py 256 20; python3 tools/hier_bench.py 4096 5) > gpurun_out/r04_g_hier_bench.txt 2>&1
cat gpurun_out/r04_g_hier_bench.txt
bash tools/profile_hier.sh r04_g 256 10 > /dev/null 2>&1
head -30 gpurun_out/r04_g_hier256_kernel_stats_by_grid.txt
cat gpurun_out/r04_g_hier256_under_rocprof.txt | tail -2
(python3 tools/strip_probe.py both 4096 10; SUHMO_TRANSPORT=ipc python3 tools/strip_probe.py strip 4096 10) 2>&1 | grep "per V-cycle" > gpurun_out/r04_strip_self_probe.txt
cat gpurun_out/r04_strip_self_probe.txt
timeout -k 10 300 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_multiproc.py -m gpu -x -q > gpurun_out/r04_g_ipc.log 2>&1; tail -2 gpurun_out/r04_g_ipc.log
