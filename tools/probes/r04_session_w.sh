cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_hier.py tests/test_gpu_hier_timestep.py tests/test_gpu_cfg5.py tests/test_gpu_hier_strips.py -m gpu -x -q > gpurun_out/r04_w_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r04_w_tests.log
[ $rc -eq 0 ] || exit $rc
python3 tools/hier_bench.py 256 20 2>&1 | grep "per step"
python3 tools/hier_bench.py 4096 5 2>&1 | grep "per step"
python3 tools/probes/amr3_via_hier.py 2>&1 | tail -3
