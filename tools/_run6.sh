mkdir -p gpurun_out/r03f
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "2p24" > gpurun_out/r03f/pytest_2p24.log 2>&1; echo "2p24 rc=$?"
timeout 1200 python bench.py --steps 2 --warmup 1 --log-constraints 24 --no-cpu-baseline --msm-micro "" --no-live-traffic > gpurun_out/r03f/bench_2p24.json 2> gpurun_out/r03f/bench_2p24.err; echo "bench 2p24 rc=$?"
PM_WIDE=0 timeout 1200 python bench.py --steps 2 --warmup 1 --log-constraints 24 --no-cpu-baseline --msm-micro "" --no-live-traffic > gpurun_out/r03f/bench_2p24_nowide.json 2> gpurun_out/r03f/bench_2p24_nowide.err; echo "bench 2p24 nowide rc=$?"
tail -n 3 gpurun_out/r03f/pytest_2p24.log
python - <<'PY'
import json
for m in ("","_nowide"):
    try:
        j=json.loads(open("gpurun_out/r03f/bench_2p24%s.json"%m).read().strip().splitlines()[-1])
        print(m, round(j["ms_per_step"],1), round(j["ms_per_step_hbm_resident"],1), j["proof_verified"], j["valu"]["mixed_adds_per_pair"], j["stage_ms_phase3"])
    except Exception as e: print(m, "ERR", e)
PY
