mkdir -p gpurun_out/r03k
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_vector.py -m gpu -x -q -k "equals_single_gpu or many_segments or mid_size" > gpurun_out/r03k/pytest_sharded.log 2>&1; echo "sharded rc=$?"
tail -n 2 gpurun_out/r03k/pytest_sharded.log
for th in 512 1024; do for lg in 13 14; do
PM_SEG_THREADS=$th PM_MAX_SEG_LOG=$lg timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03k/emu8_t${th}_s$lg.json 2>/dev/null
python -c "import json; j=json.load(open('gpurun_out/r03k/emu8_t${th}_s$lg.json')); print('threads $th seglog $lg', round(j['emulated_ms_per_rank'],3), 'poly', j['stage_ms_rank0']['poly'], j['stage_ms_last_rank']['poly'])"
done; done
