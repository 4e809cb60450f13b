# Where the NTT pass kernels spend their cycles: SQ counter passes over tools/ntt_bench.py at one size.
#   gpurun --timeout 900 -- 'bash tools/pmc_ntt.sh <tag> [log_n]'   -> gpurun_out/<tag>/
TAG=${1:-r02_ntt_pmc}
LOG=${2:-22}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o run -- python3 $GRAFT_REPO_ROOT/tools/ntt_bench.py --logs $LOG --reps 2 > $O/pmc$i.json 2> $O/pmc$i.log
  f=$(find $O/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f ntt_ | tee -a $O/summary.txt
done
