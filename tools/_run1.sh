mkdir -p gpurun_out/r03a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sharded_vector.py -m gpu -x -q -k "fails_fast or equals_single_gpu or mid_size or many_segments or world_of_one" > gpurun_out/r03a/pytest_sharded.log 2>&1; echo "sharded rc=$?" 
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench" > gpurun_out/r03a/pytest_bench.log 2>&1; echo "bench tests rc=$?"
timeout 300 python -m pytest tests/test_host_verify.py -m gpu -x -q > gpurun_out/r03a/pytest_verify.log 2>&1; echo "verify rc=$?"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --msm-micro "" --no-live-traffic > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 3 > gpurun_out/r03a/emu8.json 2> gpurun_out/r03a/emu8.err; echo "emu rc=$?"
tail -3 gpurun_out/r03a/pytest_sharded.log gpurun_out/r03a/pytest_bench.log gpurun_out/r03a/pytest_verify.log
