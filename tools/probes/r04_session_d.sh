# GPU session of round 4: strips / transports after the batch channel, hierarchy after the fused FAS enter / leave, the self-neighbour
# strip probe in both transports, the dense-level partition numbers (gloo processes on the one GPU)
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_strips.py tests/test_gpu_rccl.py tests/test_gpu_timestep_strips.py tests/test_gpu_hier.py tests/test_gpu_hier_timestep.py -m gpu -x -q > gpurun_out/r04_d_strips.log 2>&1
grep -n "ipc transport" gpurun_out/r04_d_strips.log | head -3 | cut -c1-600; tail -3 gpurun_out/r04_d_strips.log
(python3 tools/strip_probe.py both 4096 10; SUHMO_TRANSPORT=ipc python3 tools/strip_probe.py strip 4096 10) 2>&1 | grep "per V-cycle" > gpurun_out/r04_strip_self_probe.txt
cat gpurun_out/r04_strip_self_probe.txt
export GLOO_SOCKET_IFNAME=lo SUHMO_DIST_BACKEND=gloo OMP_NUM_THREADS=1
for w in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 29611 tools/hier_dist.py --base 512 --levels 3 --steps 1 --dense --partition-min-cells 1 --check 2>&1 | grep -v "^W\|warn" | tail -8
done > gpurun_out/r04_partition_dense_numbers.txt
cut -c1-900 gpurun_out/r04_partition_dense_numbers.txt
