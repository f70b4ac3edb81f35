#!/bin/bash
# round 4, session d: fused refine (survivors folded in) -- matcher tests, per-call times, bench, kernel trace
TAG=${1:-r04_d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "${KSEL:-match or planted or chain_golden or end_to_end or fuzz}" > $O/tests_new.log 2>&1; echo "tests(new) rc=$?"; tail -3 $O/tests_new.log
grep -E "FAILED|Error" $O/tests_new.log | head -20
for shape in top_l1 top_l2 top_g mid_l1 mid_g; do
  timeout 300 python tools/kbench.py match --shape $shape --data all --iters 7 2>&1 | grep -v amdgpu.ids >> $O/match_regimes.txt
done
cat $O/match_regimes.txt
timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 7 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
grep '"metric"' $O/prof.log > $O/bench_profiled.json
python $R/profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1; rm -f $O/prof/k_results.db
head -40 $O/kernel_stats.txt
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["matching"]["matching_ms_per_step"], d["roofline"]["attention_ms_per_step"], d["event_pass_ms"], d["unaccounted_ms_per_step"], d["box"]["sclk_mhz"]["mean"], d["box"]["power_w"]["mean"])
print(json.dumps(d["components_ms_per_step"]))
PY
