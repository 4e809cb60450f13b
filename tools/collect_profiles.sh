# Collects the judged evidence on an MI355X box: bench JSON lines, rocprofv3 kernel stats, the two PMC passes.
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh <tag>'   -> gpurun_out/<tag>/
set -x
TAG=${1:-r01x}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --msm-micro "" --curve bn254 > $O/bench_bn254.json 2>> $O/bench.log
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --msm-micro "" --log-constraints 22 > $O/bench_2p22.json 2>> $O/bench.log
for s in 0/2 0/4 0/8; do BENCH_FAKE_SHARD=$s timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --msm-micro "" > $O/bench_fake_shard_$(echo $s | tr / _).json 2>> $O/bench.log; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --msm-micro "" > $O/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" > /dev/null 2>&1
ls -la $O
