cd $GRAFT_REPO_ROOT
for nb in 32 64 128 192 384; do echo "SUHMO_IPC_BLOCKS=$nb"; SUHMO_IPC_BLOCKS=$nb SUHMO_TRANSPORT=ipc python3 tools/strip_probe.py strip 4096 10 2>&1 | grep "per V-cycle"; done > gpurun_out/r04_j_ipc_blocks_sweep.txt
cat gpurun_out/r04_j_ipc_blocks_sweep.txt
timeout -k 10 600 python -m pytest tests/test_gpu_hier.py tests/test_gpu_parity.py -m gpu -x -q -k "pieces or update_and_average or test_gsrb" > gpurun_out/r04_j_split.log 2>&1; tail -2 gpurun_out/r04_j_split.log
