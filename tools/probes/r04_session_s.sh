cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_rccl.py tests/test_gpu_strips.py -m gpu -x -q -k "ipc or process" > gpurun_out/r04_s_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r04_s_tests.log
exit $rc
