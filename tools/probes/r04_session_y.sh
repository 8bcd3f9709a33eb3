cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04_y_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04_y_pytest.log
python3 bench.py > gpurun_out/r04_y_bench.json 2> gpurun_out/r04_y_bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_y_bench.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["vcycle_physical_frac"]); print(json.dumps(j["converged_solve"]))
PY
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_y_bench.json") if l.startswith("{")][-1])
for k, v in j["other_configs"].items(): print(k, {a: b for a, b in v.items() if "ms" in a})
print(j.get("regression_guard"))
PY
