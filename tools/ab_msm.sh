# Same-box A/B of library builds on the standalone table-mode MSM only (sort / accumulate / reduce stage times): VARIANTS="a b" [LOGLEN=24] bash tools/ab_msm.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in ${VARIANTS:-base new}; do
  cp ab/libpolymath_hip_$v.so polymath_amd/libpolymath_hip.so
  timeout 300 python tools/msm_bench.py --tables --log-len ${LOGLEN:-24} --reps 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep best_ms %.3f' % d['best_ms'], d['stage_ms'])"
done; done
