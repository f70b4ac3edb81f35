#!/bin/bash
# round 4, session c: full GPU suite + smoke, bench at KP = 0 / 2 / 3 on one box
TAG=${1:-r04_c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
grep -E "FAILED|ERROR" $O/tests.log | head
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for kp in 0 2 3; do
  VTM_DEBUG_KP=$kp timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_kp$kp.json 2>> $O/bench.err; echo "bench kp$kp rc=$?"
done
python - <<PY
import json
for n in ("bench_kp0","bench_kp2","bench_kp3"):
    try:
        d=json.load(open("$O/%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["matching"]["matching_ms_per_step"], d["roofline"]["attention_ms_per_step"], d["event_pass_ms"], d["unaccounted_ms_per_step"], d["box"]["sclk_mhz"]["mean"], d["box"]["power_w"]["mean"])
    except Exception as e:
        print(n, "failed", e)
PY
