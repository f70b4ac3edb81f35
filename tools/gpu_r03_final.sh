#!/bin/bash
# final check of the round: full GPU suite, smoke, bench (+ CPU baseline), full-block bench, kernel trace of the bench
TAG=${1:-r03_z}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --no-cpu-baseline --full-block > $O/bench_full_block.json 2>> $O/bench.err; echo "bench full rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 7 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
grep '"metric"' $O/prof.log > $O/bench_profiled.json
python $R/profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1; rm -f $O/prof/k_results.db
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['reference_equivalent_tflops'],d['roofline']['top_block'],d['matching']['matching_ms_per_step'],d['cpu_baseline']['seconds_per_step'],d['box'])
f=json.load(open('$O/bench_full_block.json'));print(f['value'],f['ms_per_step'])"
