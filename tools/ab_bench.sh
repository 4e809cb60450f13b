# Same-box A/B of two (or more) library builds inside ONE gpurun call (boxes differ by several per cent):
#   cp polymath_amd/libpolymath_hip.so ab/libpolymath_hip_<tag>.so   for every build (variants are LOADED from ab/ via POLYMATH_HIP_LIB), then
#   gpurun --timeout 1500 -- 'VARIANTS="base new" bash tools/ab_bench.sh'
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in ${VARIANTS:-base new}; do
  export POLYMATH_HIP_LIB=$GRAFT_REPO_ROOT/ab/libpolymath_hip_$v.so      # loaded through api.LIB_PATH: the in-tree library is never overwritten
  timeout 300 python tools/msm_bench.py --tables --log-len 24 --reps 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep best_ms %.3f' % d['best_ms'], d['stage_ms'])"
done; done
for v in ${VARIANTS:-base new}; do
  export POLYMATH_HIP_LIB=$GRAFT_REPO_ROOT/ab/libpolymath_hip_$v.so      # loaded through api.LIB_PATH: the in-tree library is never overwritten
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v ms_per_step %.3f resident %s' % (d['ms_per_step'], d.get('ms_per_step_hbm_resident')), {k:round(v.get('ms'),3) for k,v in d.get('stages',{}).items() if isinstance(v,dict)})"
done
unset POLYMATH_HIP_LIB
