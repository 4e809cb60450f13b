mkdir -p gpurun_out/r03n
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_vector.py -m gpu -x -q -k "equals_single_gpu and bls" > gpurun_out/r03n/pytest.log 2>&1; echo "tests rc=$?"
PM_MSM_SPLIT_D=1 timeout 600 python -m pytest tests/test_sharded_vector.py -m gpu -x -q -k "equals_single_gpu and bls" > gpurun_out/r03n/pytest_split.log 2>&1; echo "tests split rc=$?"
for rep in 1 2; do for sp in 0 1; do
PM_MSM_SPLIT_D=$sp timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03n/emu8_split${sp}_$rep.json 2>/dev/null
python -c "import json; j=json.load(open('gpurun_out/r03n/emu8_split${sp}_$rep.json')); print('split $sp', round(j['emulated_ms_per_rank'],3), j['busy_ms_per_rank'])"
done; done
