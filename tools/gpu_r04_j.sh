#!/bin/bash
# session j: duplicate-key folding (vtm_fold_keys / vtm_attention_kv_folded) -- parity, then bench A/B
TAG=${1:-r04_j}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "${KSEL:-fold or attention or end_to_end}" > $O/tests_new.log 2>&1; echo "tests(new) rc=$?"; tail -5 $O/tests_new.log
grep -E "FAILED|Error" $O/tests_new.log | head -20
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
if [ -z "$NOBENCH" ]; then
for rep in 1 2; do
  VIDTOME_FOLD_KEYS=0 timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_nofold$rep.json 2>> $O/bench.err
  timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_fold$rep.json 2>> $O/bench.err
done
python - <<PY
import json
for n in ("nofold1","fold1","nofold2","fold2"):
    try:
        d=json.load(open("$O/bench_%s.json"%n)); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], d["matching"]["matching_ms_per_step"], r["attention_ms_per_step"], r["top_block"], r["achieved"], r.get("folded_launches"), r.get("folded_key_fraction"), d["components_ms_per_step"].get("fold_keys"))
    except Exception as e: print(n, "failed", e)
PY
tail -5 $O/bench.err
fi
