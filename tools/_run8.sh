mkdir -p gpurun_out/r03h
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for seg in 24 32 48 64 96; do
PM_MSM_SEG=$seg timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03h/emu8_msmseg$seg.json 2>/dev/null; echo "msm seg $seg rc=$?"
done
timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03h/emu8.json 2>/dev/null
for f in gpurun_out/r03h/emu8*.json; do echo $f; python -c "import json,sys; j=json.load(open('$f')); print(round(j['emulated_ms_per_rank'],3), j['busy_ms_per_rank'], j['stage_ms_rank0'])"; done
