#!/bin/bash
# usage (on an MI355X box): bash tools/pmc_sort.sh <tag> [log_len]   -- SQ counter passes over the kernels of one resident table-mode MSM
# (tools/msm_bench.py); one rocprofv3 --pmc run per counter group, no tracing.  Results: gpurun_out/<tag>/<group>/... + summary.txt
tag=${1:-pmc_sort}; ln=${2:-24}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; rocprofv3 --pmc "$@" -d $out/$name -o c --output-format csv -- python3 tools/msm_bench.py --tables --log-len $ln --reps 2 > $out/$name.log 2>&1; }
run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run wave SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
for g in valu wave lds vmem; do echo "== $g"; python3 tools/pmc_summary.py $(ls $out/$g/*counter_collection.csv | head -1) | grep -E "k_region_pass|k_tbl_partition|k_hist_small|k_tbl_count|k_accumulate|k_reduce_level0" ; done > $out/summary.txt 2>&1
cat $out/summary.txt
