# Collects the judged evidence on an MI355X box: bench JSON lines, rocprofv3 kernel stats, the two PMC traffic passes, the GPU suite.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh <tag> bench'   -> gpurun_out/<tag>/   (benches, emulations, profiler passes)
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh <tag> tests'   -> gpurun_out/<tag>/   (smoke + pytest -m gpu)
# (two calls: a gpurun call is limited to 1200 s)
set -x
TAG=${1:-r06_final}
PART=${2:-all}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
if [ "$PART" = "tests" ] || [ "$PART" = "all" ]; then
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
  tail -n 1 $O/smoke.log
  (time timeout 1100 python -m pytest tests -m gpu -x -q --durations=25) > $O/pytest_gpu.log 2>&1
  tail -n 6 $O/pytest_gpu.log
fi
if [ "$PART" = "tests" ]; then exit 0; fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --other-configs "" --curve bn254 > $O/bench_bn254.json 2>> $O/bench.log
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --other-configs "" --log-constraints 22 > $O/bench_2p22.json 2>> $O/bench.log
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --other-configs "" --log-constraints 24 > $O/bench_2p24.json 2>> $O/bench.log
for n in 2 4 8; do timeout 300 python tools/shard_emulation.py --ranks $n --steps 3 > $O/shard_emulation_vector_$n.json 2>> $O/bench.log; done
timeout 300 python tools/shard_emulation.py --ranks 8 --steps 3 --layout pairs > $O/shard_emulation_pairs_8.json 2>> $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" > $O/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" --opt msm_overlap=0 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" --opt msm_overlap=0 > /dev/null 2>&1
# where a rank-proof goes at N = 8: kernel trace of the emulation, split by rank thread
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace8 -o emu8 -- python3 $GRAFT_REPO_ROOT/tools/shard_emulation.py --ranks 8 --steps 2 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_per_rank.py $O/trace8/emu8_kernel_trace.csv > $O/kernel_stats_per_rank_proof_8ranks.csv
rm -rf $O/trace8
cd $GRAFT_REPO_ROOT
# the larger configurations as ONE proof over 8 emulated ranks (2^24 in the wide mode: eight ranks' tables do not fit one GPU)
timeout 400 python tools/shard_emulation.py --ranks 8 --steps 2 --log-constraints 22 --opt ntt_overlap=0 > $O/shard_emulation_vector_8_2p22.json 2>> $O/bench.log
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 2 --log-constraints 24 --opt ntt_overlap=0 --opt tables=wide > $O/shard_emulation_vector_8_2p24_wide.json 2>> $O/bench.log
ls -la $O
