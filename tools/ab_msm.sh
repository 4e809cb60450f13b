# Same-box A/B of library builds on the standalone table-mode MSM only (sort / accumulate / reduce stage times): VARIANTS="a b" [LOGLEN=24] bash tools/ab_msm.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in ${VARIANTS:-base new}; do
  export POLYMATH_HIP_LIB=$GRAFT_REPO_ROOT/ab/libpolymath_hip_$v.so      # loaded through api.LIB_PATH: the in-tree library is never overwritten
  timeout 300 python tools/msm_bench.py --tables --log-len ${LOGLEN:-24} --reps 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep best_ms %.3f' % d['best_ms'], d['stage_ms'])"
done; done
unset POLYMATH_HIP_LIB
