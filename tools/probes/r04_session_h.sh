cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_hier.py tests/test_gpu_hier_timestep.py tests/test_gpu_cfg5.py -m gpu -x -q -k "not north_star" > gpurun_out/r04_h_hier.log 2>&1
tail -3 gpurun_out/r04_h_hier.log
(python3 tools/hier_bench.py 256 20; python3 tools/hier_bench.py 4096 5) > gpurun_out/r04_h_hier_bench.txt 2>&1
cat gpurun_out/r04_h_hier_bench.txt
bash tools/profile_hier.sh r04_h 256 10 > /dev/null 2>&1
head -12 gpurun_out/r04_h_hier256_kernel_stats_by_grid.txt
export TMPDIR=/tmp; cd /tmp && SUHMO_TRANSPORT=ipc rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_h_ipc_prof -- python3 $GRAFT_REPO_ROOT/tools/strip_probe.py strip 4096 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/stats_by_grid.py $(ls gpurun_out/r04_h_ipc_prof/*/*kernel_trace.csv | head -1) 13 > gpurun_out/r04_h_ipc_strip_kernel_stats_by_grid.txt; grep -n "ipc\|rccl\|pack" gpurun_out/r04_h_ipc_strip_kernel_stats_by_grid.txt | head
rm -rf gpurun_out/r04_h_ipc_prof
