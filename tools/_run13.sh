mkdir -p gpurun_out/r03m
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for q in 4 8 24; do
GPU_MAX_HW_QUEUES=$q timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03m/emu8_q$q.json 2>/dev/null
python -c "import json; j=json.load(open('gpurun_out/r03m/emu8_q$q.json')); print('queues $q', round(j['emulated_ms_per_rank'],3), j['busy_ms_per_rank'], j['stage_ms_rank0']['msm_total'])"
done
GPU_MAX_HW_QUEUES=24 PM_MSM_OVERLAP=0 timeout 600 python tools/shard_emulation.py --ranks 8 --steps 4 > gpurun_out/r03m/emu8_q24_ov0.json 2>/dev/null
python -c "import json; j=json.load(open('gpurun_out/r03m/emu8_q24_ov0.json')); print('queues 24 overlap 0', round(j['emulated_ms_per_rank'],3), j['busy_ms_per_rank'])"
