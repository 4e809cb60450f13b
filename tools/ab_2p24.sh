# Same-box A/B of library builds on the 2^24 - 100-gate proof on one GPU (wide-mode MSMs): VARIANTS="regions512 regions1024" bash tools/ab_2p24.sh
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-regions512 regions1024}; do
  export POLYMATH_HIP_LIB=$GRAFT_REPO_ROOT/ab/libpolymath_hip_$v.so      # loaded through api.LIB_PATH: the in-tree library is never overwritten
  timeout 900 python bench.py --steps 3 --warmup 1 --log-constraints 24 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v ms_per_step %.2f resident %.2f verified %s adds/pair %s' % (d['ms_per_step'], d['ms_per_step_hbm_resident'], d['proof_verified'], d['valu']['mixed_adds_per_pair']), {k:round(v.get('ms'),2) for k,v in d.get('stages',{}).items() if isinstance(v,dict)}, d['proof_bytes'][:32])"
done
unset POLYMATH_HIP_LIB
