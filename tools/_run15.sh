mkdir -p gpurun_out/r03o
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench or skew" > gpurun_out/r03o/pytest.log 2>&1; echo "tests rc=$?"
tail -n 30 gpurun_out/r03o/pytest.log
