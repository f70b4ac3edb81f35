#!/bin/bash
# session h: seeded matcher -- parity, per-call times (seeded through compute_merge only: bench A/B), regimes
TAG=${1:-r04_h}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chunk_parallel.py -m gpu -q -k "${KSEL:-match or planted or chain or end_to_end or fuzz or cfg or live or full_size or seeds}" > $O/tests_new.log 2>&1; echo "tests(new) rc=$?"; tail -3 $O/tests_new.log
grep -E "FAILED|Error" $O/tests_new.log | head -20
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for rep in 1 2; do
  VIDTOME_SEED=0 timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_noseed$rep.json 2>> $O/bench.err
  timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_seed$rep.json 2>> $O/bench.err
done
for d in n01 corr01 flat25 dup; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --data $d > $O/bench_$d.json 2>> $O/bench.err; echo "bench $d rc=$?"
done
python - <<PY
import json
for n in ("bench_noseed1","bench_seed1","bench_noseed2","bench_seed2","bench_n01","bench_corr01","bench_flat25","bench_dup"):
    try:
        d=json.load(open("$O/%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["matching"]["matching_ms_per_step"], d["matching"]["counters"], d["roofline"]["attention_ms_per_step"], d["box"]["sclk_mhz"]["mean"])
    except Exception as e: print(n, "failed", e)
PY
