cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_hier.py tests/test_gpu_hier_timestep.py tests/test_gpu_cfg5.py tests/test_gpu_hier_strips.py -m gpu -x -q > gpurun_out/r04_o_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r04_o_tests.log
[ $rc -eq 0 ] || exit $rc
python3 tools/hier_bench.py 256 20 > gpurun_out/r04_o_hier256.txt 2>&1; cat gpurun_out/r04_o_hier256.txt
python3 tools/hier_bench.py 4096 5 > gpurun_out/r04_o_hier4096.txt 2>&1; cat gpurun_out/r04_o_hier4096.txt
