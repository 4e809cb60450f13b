# L2 hit rate and fabric requests of k_accumulate: MI355X_MICROARCH.md's recipe TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum),
# TCC_EA0_RDREQ_sum beside FETCH_SIZE, per library variant in ab/.  One counter set per pass, kernel trace only.
#   gpurun --timeout 1500 -- 'VARIANTS="idx1 idx4" bash tools/pmc_l2.sh <tag>'   -> gpurun_out/<tag>/
TAG=${1:-r06_pmc_l2}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-idx1 idx4}; do
  export POLYMATH_HIP_LIB=$GRAFT_REPO_ROOT/ab/libpolymath_hip_$v.so
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${v}_pmc$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --msm-micro "" --ntt-micro "" --no-live-traffic --inflight 0 --other-configs "" --opt msm_overlap=0 > $O/${v}_pmc$i.json 2> $O/${v}_pmc$i.log
    f=$(find $O/${v}_pmc$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - "$f" "$v" <<'PY' | tee -a $O/summary.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_accumulate" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, vals in sorted(by.items()):
    print("%s k_accumulate %s: %d launches, per launch %s, mean %.6g" % (sys.argv[2], name, len(vals), ["%.6g" % v for v in vals], sum(vals) / len(vals)))
PY
    rm -rf $O/${v}_pmc$i
  done
done
unset POLYMATH_HIP_LIB
