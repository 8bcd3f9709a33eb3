cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04_z_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04_z_pytest.log
bash tools/profile_round.sh r04_z > gpurun_out/r04_z_profile_round.log 2>&1; echo "profile_round rc $?"; tail -2 gpurun_out/r04_z_profile_round.log
bash tools/pmc_vcycle.sh r04_z > gpurun_out/r04_z_pmc_vcycle.log 2>&1; echo "pmc_vcycle rc $?"; tail -2 gpurun_out/r04_z_pmc_vcycle.log
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_z_bench.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["vcycle_physical_frac"], j["roofline"]["physical_frac"])
for k, v in j["other_configs"].items(): print(k, {a: b for a, b in v.items() if "ms" in a})
print(j.get("regression_guard"))
print(open("gpurun_out/r04_z_pmc_traffic_vcycle.json").read()[:600])
PY
rm -rf gpurun_out/r04_z_prof gpurun_out/r04_z_pmcv gpurun_out/r04_z_pmc
