mkdir -p gpurun_out/r03g
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lg in 12 13 14 15; do
PM_MAX_SEG_LOG=$lg timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03g/emu8_seg$lg.json 2>/dev/null; echo "seg $lg rc=$?"
done
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03g/emu8.json 2>/dev/null
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03g/trace -o emu8 -- python3 $GRAFT_REPO_ROOT/tools/shard_emulation.py --ranks 8 --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r03g/emu8_traced.json 2> /dev/null; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_per_rank.py gpurun_out/r03g/trace/emu8_kernel_trace.csv > gpurun_out/r03g/kernel_stats_per_rank_proof_8ranks.csv
rm -rf gpurun_out/r03g/trace
for f in gpurun_out/r03g/emu8*.json; do echo $f; python -c "import json,sys; j=json.load(open('$f')); print(round(j['emulated_ms_per_rank'],3), j['busy_ms_per_rank'])"; done
head -12 gpurun_out/r03g/kernel_stats_per_rank_proof_8ranks.csv; tail -4 gpurun_out/r03g/kernel_stats_per_rank_proof_8ranks.csv
