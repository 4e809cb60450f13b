mkdir -p gpurun_out/r03c
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03c/trace -o emu8 -- python3 $GRAFT_REPO_ROOT/tools/shard_emulation.py --ranks 8 --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r03c/emu8_traced.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03c/emu8_traced.err; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/r03c/trace/* | head
# wide-mode MSM correctness + timing at 2^18 and 2^20 circuits against the table mode
for mode in 1 wide 0; do
PM_TABLES=$mode timeout 600 python bench.py --steps 3 --warmup 1 --log-constraints 20 --no-cpu-baseline --msm-micro "" --no-live-traffic > gpurun_out/r03c/bench_tables_$mode.json 2> gpurun_out/r03c/bench_tables_$mode.err; echo "bench tables=$mode rc=$?"
done
python - <<'PY'
import json
for m in ("1","wide","0"):
    try:
        j=json.loads(open("gpurun_out/r03c/bench_tables_%s.json"%m).read().strip().splitlines()[-1])
        print(m, round(j["ms_per_step"],2), j["proof_verified"], j["valu"]["mixed_adds_per_pair"], j["proof_bytes"][:24], j["stage_ms_phase3"])
    except Exception as e: print(m, "ERR", e)
PY
