#!/bin/bash
# round 3 GPU session: full GPU suite, smoke, bench (both regimes) + kernel trace
TAG=${1:-r03_a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --same-chunk > $O/bench_same_chunk.json 2> $O/bench_same.err; echo "bench same rc=$?"
timeout 300 python bench.py --no-cpu-baseline --exchange neighbour > $O/bench_n1_neighbour.json 2> $O/bench_nb.err; echo "bench nb rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 7 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
grep '"metric"' $O/prof.log > $O/bench_profiled.json
cd $R
python profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1
head -24 $O/kernel_stats.txt | cut -c1-70,100-170
python - <<PY
import json
for n in ("bench.json","bench_same_chunk.json","bench_n1_neighbour.json"):
    try:
        d=json.load(open("$O/"+n)); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["top_block"], d["matching"]["matching_ms_per_step"], d.get("box"))
    except Exception as e: print(n, "ERR", e)
PY
