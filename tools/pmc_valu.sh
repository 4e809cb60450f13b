# VALU evidence for k_accumulate: SQ counter passes over one bench run, csv per dispatch.
#   gpurun --timeout 1500 -- 'bash tools/pmc_valu.sh <tag>'   -> gpurun_out/<tag>/
set -x
TAG=${1:-r02_valu}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq1 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" --opt msm_overlap=0 > $O/pmc_sq1.json 2> $O/pmc_sq1.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" --opt msm_overlap=0 > $O/pmc_sq2.json 2> $O/pmc_sq2.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq3 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" --opt msm_overlap=0 > $O/pmc_sq3.json 2> $O/pmc_sq3.log
find $O -name '*.csv' | xargs ls -la
