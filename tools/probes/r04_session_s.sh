cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py -m gpu -x -q > gpurun_out/r04_s_tests.log 2>&1; rc=$?; tail -25 gpurun_out/r04_s_tests.log
exit $rc
