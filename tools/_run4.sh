mkdir -p gpurun_out/r03d
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sharded_vector.py -m gpu -x -q -k "not 2p22 and not 2p24" > gpurun_out/r03d/pytest_sharded.log 2>&1; echo "sharded rc=$?"
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "wide or piece_split or bit_exact" > gpurun_out/r03d/pytest_configs.log 2>&1; echo "configs rc=$?"
for ov in 1 0; do
PM_MSM_OVERLAP=$ov timeout 600 python tools/shard_emulation.py --ranks 8 --steps 3 > gpurun_out/r03d/emu8_ov$ov.json 2> gpurun_out/r03d/emu8_ov$ov.err; echo "emu ov=$ov rc=$?"
done
timeout 600 python tools/shard_emulation.py --ranks 4 --steps 3 > gpurun_out/r03d/emu4.json 2>/dev/null
timeout 600 python tools/shard_emulation.py --ranks 2 --steps 3 > gpurun_out/r03d/emu2.json 2>/dev/null
tail -n 3 gpurun_out/r03d/pytest_sharded.log gpurun_out/r03d/pytest_configs.log
for f in gpurun_out/r03d/emu*.json; do echo $f; cut -c1-330 $f; echo; done
