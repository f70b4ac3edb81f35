#!/bin/bash
# End-of-round record: full GPU suite, smoke, bench (+ CPU leg), two more bench runs (spread), kernel trace of the bench.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r04_end.sh r04_z'
TAG=${1:-r04_z}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
grep -E "FAILED|ERROR" $O/tests.log | head
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for rep in 2 3; do
  timeout 600 python bench.py --no-cpu-baseline > $O/bench_run$rep.json 2>> $O/bench.err
done
timeout 600 python bench.py --no-cpu-baseline --full-block > $O/bench_full_block.json 2>> $O/bench.err; echo "bench full rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 7 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
grep '"metric"' $O/prof.log > $O/bench_profiled.json
python $R/profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1; rm -rf $O/prof
python - <<PY
import json
for n in ("bench","bench_run2","bench_run3","bench_full_block","bench_profiled"):
    try:
        d=json.load(open("$O/%s.json"%n)); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["frac"], r["frac_at_sustained_clock"], r["top_block"], d["matching"]["matching_ms_per_step"], r["attention_ms_per_step"], d["box"]["sclk_mhz"]["mean"], d.get("cpu_baseline",{}).get("seconds_per_step"))
    except Exception as e: print(n, "failed", e)
PY
head -12 $O/kernel_stats.txt | cut -c1-150
