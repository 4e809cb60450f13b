# the remaining cheap levers of the accumulate path, measured on the quotient MSM's shape
# (20 971 542 pairs, window tables): 11 windows (c = 24, 2^23 buckets) against the planned 12, and the task length SEG.
#   gpurun --timeout 900 -- 'bash tools/sweep_levers.sh'  -> gpurun_out/r02_levers.jsonl
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_levers.jsonl
: > $O
L=20971542
timeout 120 python tools/msm_bench.py --len $L --tables --reps 3 >> $O
PM_TABLE_C=24 timeout 120 python tools/msm_bench.py --len $L --tables --reps 3 >> $O
for seg in 128 192 256 384 512; do PM_MSM_SEG=$seg timeout 120 python tools/msm_bench.py --len $L --tables --reps 3 >> $O; done
for k0 in 8 16 32; do PM_RED_K0=$k0 timeout 120 python tools/msm_bench.py --len $L --tables --reps 3 >> $O; done
# the small MSMs of an 8-GPU shard (2.6 M and 0.8 M pairs): level-0 fan-in of the reduction
for len in 2621440 786432 262144; do for k0 in 2 4 8; do PM_RED_K0=$k0 timeout 120 python tools/msm_bench.py --len $len --tables --reps 5 >> $O; done; done
cat $O | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['len'], d['env'], round(d['best_ms'], 3), d['stage_ms'])"
