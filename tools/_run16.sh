mkdir -p gpurun_out/r03p
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_sharded_vector.py -m gpu -x -q > gpurun_out/r03p/pytest_sharded.log 2>&1; echo "sharded rc=$?"
tail -n 12 gpurun_out/r03p/pytest_sharded.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_two_ranks or sharded" > gpurun_out/r03p/pytest_bench.log 2>&1; echo "bench rc=$?"
for rep in 1 2; do
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03p/emu8_$rep.json 2>/dev/null
python -c "import json; j=json.load(open('gpurun_out/r03p/emu8_$rep.json')); print(round(j['emulated_ms_per_rank'],3), j['busy_ms_per_rank'], j['stage_ms_rank0'])"
done
PM_PROFILE_HOST=1 timeout 600 python tools/shard_emulation.py --ranks 8 --steps 2 2>&1 >/dev/null | grep "rank 0" | tail -3
