cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python3 bench.py ) > gpurun_out/r04_k_bench.json 2> gpurun_out/r04_k_bench.err; echo "bench rc $?"; tail -4 gpurun_out/r04_k_bench.err
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_k_bench.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["physical_frac"], j["roofline"]["vcycle_hbm_bytes"], j["roofline"]["vcycle_physical_frac"])
print(json.dumps(j["converged_solve"])); print(json.dumps(j["regression_guard"]))
print({k: (v.get("ms_per_vcycle") or v.get("ms_per_step")) for k, v in j["other_configs"].items()})
PY
bash tools/profile_hier.sh r04_k 256 10 > /dev/null 2>&1; head -8 gpurun_out/r04_k_hier256_kernel_stats_by_grid.txt
