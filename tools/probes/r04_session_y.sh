cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04_y_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04_y_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
