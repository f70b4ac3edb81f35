#!/bin/bash
# round 4, session a: the matcher's escape (parity + time), data regimes, parity pins, smoke, bench
TAG=${1:-r04_a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "${KSEL:-match_filtered or planted or default_fp16_path or device_generator or chain_golden}" > $O/tests_new.log 2>&1; echo "tests(new) rc=$?"; tail -5 $O/tests_new.log
grep -E "filtered ms / exact ms|worst block-output" $O/tests_new.log
timeout 300 python -m pytest tests/test_gpu_chunk_parallel.py -m gpu -q -x -k "cfg4" > $O/tests_cfg4.log 2>&1; echo "tests(cfg4) rc=$?"; tail -3 $O/tests_cfg4.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for shape in top_l1 top_l2 top_g mid_l1 mid_g; do
  timeout 300 python tools/kbench.py match --shape $shape --data all --iters 5 >> $O/match_regimes.txt 2>&1
done
cat $O/match_regimes.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for d in n01 corr01 flat25 dup; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --data $d > $O/bench_$d.json 2>> $O/bench.err; echo "bench $d rc=$?"
done
python - <<PY
import json
for n in ("bench","bench_n01","bench_corr01","bench_flat25","bench_dup"):
    try:
        d=json.load(open("$O/%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_at_sustained_clock"), d["matching"]["matching_ms_per_step"], d["matching"]["counters"], d["roofline"]["attention_ms_per_step"], d["unaccounted_ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
d=json.load(open("$O/bench.json")); print(json.dumps(d["components_ms_per_step"])); print(d["box"])
PY
