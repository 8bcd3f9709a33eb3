cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04_i_pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04_i_pytest.log
(python3 tools/strip_probe.py both 4096 10; SUHMO_TRANSPORT=ipc python3 tools/strip_probe.py strip 4096 10) 2>&1 | grep "per V-cycle" > gpurun_out/r04_strip_self_probe.txt
cat gpurun_out/r04_strip_self_probe.txt
