cd $GRAFT_REPO_ROOT
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_multiproc.py tests/test_gpu_strips.py -m gpu -x -q -k "ipc or process" > gpurun_out/r04_m_ipc_tests.log 2>&1 || { tail -30 gpurun_out/r04_m_ipc_tests.log; exit 1; }
tail -2 gpurun_out/r04_m_ipc_tests.log
{ python3 tools/strip_probe.py whole 4096 20 2>&1 | grep "per V-cycle"
  python3 tools/strip_probe.py strip 4096 20 2>&1 | grep "per V-cycle"
  SUHMO_TRANSPORT=ipc python3 tools/strip_probe.py strip 4096 20 2>&1 | grep "per V-cycle"
  SUHMO_TRANSPORT=ipc python3 tools/strip_probe.py strip 4096 20 2>&1 | grep "per V-cycle"; } > gpurun_out/r04_m_strip_self_probe.txt
cat gpurun_out/r04_m_strip_self_probe.txt
cd /tmp && export TMPDIR=/tmp
SUHMO_TRANSPORT=ipc timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_m_prof -o ipc -- python3 $GRAFT_REPO_ROOT/tools/strip_probe.py strip 4096 10 > $GRAFT_REPO_ROOT/gpurun_out/r04_m_prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/r04_m_prof | head
