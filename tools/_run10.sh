mkdir -p gpurun_out/r03j
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03j/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
tail -n 25 gpurun_out/r03j/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r03j/bench_default.json 2> gpurun_out/r03j/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03j/bench_default.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step","ms_per_step_hbm_resident","proof_verified")}, j["roofline"]["frac"], j["roofline"]["traffic"], j["cpu_baseline"]["value"], [ (m["len"], round(m["ms"],2)) for m in j["msm_micro"]])
PY
