mkdir -p gpurun_out/r03b
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sharded_vector.py -m gpu -x -q -k "not 2p22 and not 2p24" > gpurun_out/r03b/pytest_sharded.log 2>&1; echo "sharded rc=$?" 
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench or sharded" > gpurun_out/r03b/pytest_bench.log 2>&1; echo "bench tests rc=$?"
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 3 > gpurun_out/r03b/emu8.json 2> gpurun_out/r03b/emu8.err; echo "emu rc=$?"
timeout 600 python tools/shard_emulation.py --ranks 2 --steps 3 > gpurun_out/r03b/emu2.json 2> gpurun_out/r03b/emu2.err; echo "emu2 rc=$?"
tail -3 gpurun_out/r03b/pytest_sharded.log gpurun_out/r03b/pytest_bench.log
cut -c1-700 gpurun_out/r03b/emu8.json
