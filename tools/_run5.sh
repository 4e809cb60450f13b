mkdir -p gpurun_out/r03e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "wide or piece_split or bit_exact" > gpurun_out/r03e/pytest_configs.log 2>&1; echo "configs rc=$?"
PM_PROFILE_HOST=1 timeout 600 python tools/shard_emulation.py --ranks 8 --steps 2 > gpurun_out/r03e/emu8_prof.json 2> gpurun_out/r03e/emu8_prof.err; echo "emu prof rc=$?"
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03e/emu8.json 2> gpurun_out/r03e/emu8.err; echo "emu rc=$?"
tail -n 3 gpurun_out/r03e/pytest_configs.log
cut -c1-330 gpurun_out/r03e/emu8.json; echo
grep "pm host rank 0" gpurun_out/r03e/emu8_prof.err | tail -6
