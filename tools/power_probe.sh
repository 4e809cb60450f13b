#!/bin/bash
# usage (on an MI355X box): bash tools/power_probe.sh  -- samples rocm-smi (power, sclk, temperature) while a 2^24-pair table-mode MSM runs
# in a loop (k_accumulate is > 85 % of it), and once idle: is the accumulation clock-throttled by the power limit?
cd $GRAFT_REPO_ROOT
echo "== idle"; rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>&1 | grep -E "Power|sclk|Temperature \(Sensor (edge|junction)|Max Graphics" | head -8
python3 tools/msm_bench.py --tables --log-len 24 --reps 900 > /tmp/probe_msm.json 2>/dev/null &
pid=$!
sleep 12
for i in 1 2 3 4 5; do echo "== under load, sample $i"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|Temperature \(Sensor (edge|junction)" | head -6; sleep 1; done
wait $pid
python3 -c "import json;d=json.load(open('/tmp/probe_msm.json'));print('best_ms',d['best_ms'],d['stage_ms'])"
