# Same-box A/B of library builds on the standalone transforms: VARIANTS="base new" bash tools/ab_ntt.sh   (inside ONE gpurun call)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in ${VARIANTS:-base new}; do
  export POLYMATH_HIP_LIB=$GRAFT_REPO_ROOT/ab/libpolymath_hip_$v.so      # loaded through api.LIB_PATH: the in-tree library is never overwritten
  timeout 300 python tools/ntt_bench.py --logs ${LOGS:-18,20,21,22,24} --reps 9 2>/dev/null | python -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$v rep $rep', ' '.join('2^%d fwd %.4f inv %.4f %s' % (r['log_n'], r['fwd_ms'], r['inv_ms'], 'ok' if r['round_trips_ok'] else 'BAD') for r in rows))"
done; done
unset POLYMATH_HIP_LIB
