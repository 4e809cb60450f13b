for len in 20971542 2621440; do
for k0 in 4 8 16 32; do for r in 1 2 4 8; do
echo -n "len $len K0 $k0 R $r: "; PM_RED_K0=$k0 PM_RED_R=$r timeout 300 python tools/msm_bench.py --len $len --reps 3 --tables 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['best_ms'],2), d['stage_ms'])"
done; done; done
