mkdir -p gpurun_out/r03l
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_vector.py -m gpu -x -q -k "watchdog or world_of_one" > gpurun_out/r03l/pytest_watchdog.log 2>&1; echo "watchdog rc=$?"
tail -n 30 gpurun_out/r03l/pytest_watchdog.log
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03l/emu8.json 2>/dev/null
python -c "import json; j=json.load(open('gpurun_out/r03l/emu8.json')); print(round(j['emulated_ms_per_rank'],3), j['busy_ms_per_rank'], j['stage_ms_rank0'])"
