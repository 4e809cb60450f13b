mkdir -p gpurun_out/r03i
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# the fallback chain on a box where RCCL cannot serve two ranks (one GPU): how long does each attempt take to fail, which one wins?
( time BENCH_FORCE_DEVICE=0 BENCH_ATTEMPT_DEADLINE_S=300 timeout 1500 python bench.py --gpus 2 --steps 2 --warmup 1 --log-constraints 12 --no-cpu-baseline --msm-micro "" --no-live-traffic > gpurun_out/r03i/chain.json 2> gpurun_out/r03i/chain.err ) 2> gpurun_out/r03i/chain.time; echo "chain rc=$?"
grep -a "launch\]\|attempt\|pm_comm\|Error\|error" gpurun_out/r03i/chain.err | head -40
cat gpurun_out/r03i/chain.time
python - <<'PY'
import json
try:
    j=json.loads(open("gpurun_out/r03i/chain.json").read().strip().splitlines()[-1]); print(j["n_gpus"], j["n_ranks_seen"], j["exchange"], j["launch"])
except Exception as e: print("ERR", e)
PY
