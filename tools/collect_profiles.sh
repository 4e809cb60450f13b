# Collects the judged evidence on an MI355X box: bench JSON lines, rocprofv3 kernel stats, the two PMC traffic passes.
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh <tag>'   -> gpurun_out/<tag>/
set -x
TAG=${1:-r05_final}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
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
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -n 3 $O/pytest_gpu.log
ls -la $O
