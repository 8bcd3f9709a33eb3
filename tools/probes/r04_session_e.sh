cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_hier.py tests/test_gpu_hier_timestep.py tests/test_gpu_hier_strips.py tests/test_gpu_cfg5.py tests/test_gpu_checkpoint.py -m gpu -x -q -k "not north_star" > gpurun_out/r04_e_hier.log 2>&1
tail -4 gpurun_out/r04_e_hier.log
for rep in 1 2; do timeout -k 10 300 python -m pytest tests/test_gpu_strips.py -m gpu -x -q -k "ipc" > gpurun_out/r04_e_ipc_$rep.log 2>&1; grep -n "ipc transport" gpurun_out/r04_e_ipc_$rep.log | head -2 | cut -c1-500; tail -2 gpurun_out/r04_e_ipc_$rep.log; done
(python3 tools/strip_probe.py both 4096 10; SUHMO_TRANSPORT=ipc python3 tools/strip_probe.py strip 4096 10) 2>&1 | grep "per V-cycle" > gpurun_out/r04_strip_self_probe.txt
cat gpurun_out/r04_strip_self_probe.txt
(python3 tools/hier_bench.py 256 20; python3 tools/hier_bench.py 4096 5) > gpurun_out/r04_e_hier_bench.txt 2>&1
cat gpurun_out/r04_e_hier_bench.txt
