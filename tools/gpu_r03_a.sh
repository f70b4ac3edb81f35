#!/bin/bash
# round 3, session a: full GPU suite, bench in both regimes (+ kernel traces), regime / duplicate-query analysis
TAG=${1:-r03_a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
ls /sys/class/drm/ > $O/sysfs.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $O/sysfs.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --same-chunk > $O/bench_same_chunk.json 2> $O/bench_same.err; echo "bench same rc=$?"
timeout 300 python tools/regime.py --passes 7 > $O/regime_rotate.jsonl 2> $O/regime.err; echo "regime rc=$?"
timeout 300 python tools/regime.py --passes 4 --same-chunk > $O/regime_same.jsonl 2>> $O/regime.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_same -o k -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --same-chunk > $O/prof_same.log 2>&1; echo "prof same rc=$?"
grep '"metric"' $O/prof.log > $O/bench_profiled.json
ls $O $O/prof | head -40
python - <<PY
import json
for n in ("bench.json","bench_same_chunk.json"):
    try:
        d=json.load(open("$O/"+n)); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["top_block"], d["matching"], d.get("box"))
    except Exception as e: print(n, "ERR", e)
PY
cd $R
python profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1
python profiles/summarize_rocpd.py $O/prof_same/k_results.db > $O/kernel_stats_same.txt 2>&1
head -30 $O/kernel_stats.txt
rm -f $O/prof/k_results.db $O/prof_same/k_results.db
